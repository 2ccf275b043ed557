"""The three 1x1 stride-2 shortcut layers of the ResNet (forward, data gradient into an existing tensor, weight
gradient) under WGRAD_WGS settings."""
import torch
from asvspoof2021_air_amd import ops, _hip
def timeit(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 64
CFG = {"l2.0.sc": (64, 18, 750, 128), "l3.0.sc": (128, 9, 375, 256), "l4.0.sc": (256, 5, 188, 512)}
for name, (Cin, H, W, Cout) in CFG.items():
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05
    y = ops.conv2d_fwd(x, w, 2, 0); dy = torch.randn_like(y); dx = torch.zeros_like(x)
    dw0 = ops.conv2d_wgrad(x, dy, w.shape, 2, 0)
    with _hip.options(CONV_S2=0):
        tf0 = timeit(lambda: ops.conv2d_fwd(x, w, 2, 0))
    y1 = ops.conv2d_fwd(x, w, 2, 0)
    print("   fwd 32-channel chunks %.1f us, diff %.1e" % (tf0, ((y1 - y).abs().max() / y.abs().max()).item()))
    tf = timeit(lambda: ops.conv2d_fwd(x, w, 2, 0))
    td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 2, 0, accumulate=dx, out=dx))
    line = "%-8s fwd %.1f us | dgrad %.1f us | wgrad" % (name, tf, td)
    for wgs in (128, 256, 384):  # (doubled inside the library for these layers)
        with _hip.options(WGRAD_WGS=wgs):
            dw = ops.conv2d_wgrad(x, dy, w.shape, 2, 0)
            err = ((dw - dw0).abs().max() / dw0.abs().max()).item()
            line += " %d: %.1f us (d %.0e)" % (wgs, timeit(lambda: ops.conv2d_wgrad(x, dy, w.shape, 2, 0)), err)
    print(line, flush=True)
