"""A/B of the stride-2 3x3 direct kernels (forward, weight gradient) at the ResNet's shapes under option sets.
usage: python tools/kbench_s2.py [B] [reps]"""
import sys, torch
from asvspoof2021_air_amd import ops, _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CFG = {"l2s": (64, 18, 750, 128), "l3s": (128, 9, 375, 256), "l4s": (256, 5, 188, 512)}
SETS = [("base", {"CONV_S2": 0}), ("fwd ck4", {"CONV_S2": 1}), ("fwd ck4 mt1", {"CONV_S2": 1, "CONV_MT": 1}), ("wg 512", {"WGRAD_WGS": 512})]
def timeit(f, n=reps):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, (Cin, H, W, Cout) in CFG.items():
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    y0 = ops.conv2d_fwd(x, w, 2, 1); dy = torch.randn_like(y0)
    dw0 = ops.conv2d_wgrad(x, dy, w.shape, 2, 1)
    fl = 2.0 * y0.numel() * Cin * 9
    for label, opts in SETS:
        with _hip.options(**opts):
            y = ops.conv2d_fwd(x, w, 2, 1); dw = ops.conv2d_wgrad(x, dy, w.shape, 2, 1)
            ey = ((y - y0).abs().max() / y0.abs().max()).item(); ew = ((dw - dw0).abs().max() / dw0.abs().max()).item()
            tf = timeit(lambda: ops.conv2d_fwd(x, w, 2, 1)); tw = timeit(lambda: ops.conv2d_wgrad(x, dy, w.shape, 2, 1))
        print("%-4s %-12s fwd %.3f ms %5.1f TF (d %.1e) | wgrad %.3f ms %5.1f TF (d %.1e)" % (
            name, label, tf, fl / tf / 1e9, ey, tw, fl / tw / 1e9, ew), flush=True)
