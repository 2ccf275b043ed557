"""Times the 16 -> 64 1x1 weight gradient (ResNet layer1.0 shortcut) at B = 64, 18 x 750, both kernels."""
import torch
from asvspoof2021_air_amd import _hip, ops


def timeit(fn, reps=30):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


x = torch.randn(64, 16, 18, 750, device="cuda")
dy = torch.randn(64, 64, 18, 750, device="cuda")
sc, sh = torch.rand(16, device="cuda") + 0.5, torch.randn(16, device="cuda") * 0.1
for opt in (1, 0):
    with _hip.options(SKINNY_WGRAD=opt):
        t0 = timeit(lambda: ops.conv2d_wgrad(x, dy, (64, 16, 1, 1), 1, 0))
        t1 = timeit(lambda: ops.conv2d_wgrad(x, dy, (64, 16, 1, 1), 1, 0, sc, sh, relu=True))
    print("SKINNY_WGRAD=%d: plain %.1f us, with BN+ReLU prologue %.1f us (276 MB of operands)" % (opt, t0, t1))
for opt in (1, 0):
    with _hip.options(SKINNY_WGRAD=opt):
        t0 = timeit(lambda: ops.conv2d_wgrad(x, dy, (64, 16, 3, 3), 1, 1))
    print("3x3 SKINNY_WGRAD=%d: %.1f us (15.9 GFLOP)" % (opt, t0))
