"""Micro-benchmark of the 3x3 / stride 1 Winograd forward (+residual) and dgrad at the ResNet's shapes."""
import sys, torch
from asvspoof2021_air_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None     # e.g. l2 or l1,l4
passes = sys.argv[4] if len(sys.argv) > 4 else "frdw"             # f fwd, r fwd+res, d dgrad, w wgrad
CFG = {"l10": (16, 18, 750, 64), "l1": (64, 18, 750, 64), "l2": (128, 9, 375, 128), "l3": (256, 5, 188, 256), "l4": (512, 3, 94, 512)}
def timeit(f, n=reps):
    for _ in range(max(2, n)): f()   # as many warm-up launches as timed ones: the clocks ramp over tens of ms
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, (Cin, H, W, Cout) in CFG.items():
    if only and name not in only: continue
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    y = ops.conv2d_fwd(x, w, 1, 1); res = torch.randn_like(y); dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * 9
    nan = float("nan")
    tf = timeit(lambda: ops.conv2d_fwd(x, w, 1, 1)) if "f" in passes else nan
    tr = timeit(lambda: ops.conv2d_fwd(x, w, 1, 1, residual=res)) if "r" in passes else nan
    td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1)) if "d" in passes else nan
    tw = timeit(lambda: ops.conv2d_wgrad(x, dy, w.shape, 1, 1)) if "w" in passes else nan
    print("%-4s B=%d  fwd %.3f ms %.1f TF | fwd+res %.3f ms %.1f TF | dgrad %.3f ms %.1f TF | wgrad %.3f ms %.1f TF" % (
        name, B, tf, fl / tf / 1e9, tr, fl / tr / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9), flush=True)
