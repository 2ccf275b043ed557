"""Times the OC-Softmax head (air_ocsoftmax_fwd / _bwd) at the two batch shapes of the bench."""
import torch
from asvspoof2021_air_amd import ops


def t(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B in (64, 128):
    x = torch.randn(B, 256, device="cuda")
    c = torch.randn(1, 256, device="cuda")
    lab = (torch.arange(B, device="cuda") % 2).long()
    print("B = %3d: fwd %.1f us, bwd %.1f us (python call + launch included)" % (
        B, t(lambda: ops.ocsoftmax_fwd(x, c, lab, 0.9, 0.2, 20.0)), t(lambda: ops.ocsoftmax_bwd(x, c, lab, 0.9, 0.2, 20.0))))
