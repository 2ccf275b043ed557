"""Micro-benchmark of single conv configurations (fwd / dgrad / wgrad) at the ResNet's real shapes."""
import sys, torch
from asvspoof2021_air_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "all"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
CFG = {  # name: (Cin, H, W, Cout, k, s, p)
    "l1": (64, 18, 750, 64, 3, 1, 1), "l2": (128, 9, 375, 128, 3, 1, 1), "l3": (256, 5, 188, 256, 3, 1, 1),
    "l4": (512, 3, 94, 512, 3, 1, 1), "l2s": (64, 18, 750, 128, 3, 2, 1), "l3s": (128, 9, 375, 256, 3, 2, 1),
    "l4s": (256, 5, 188, 512, 3, 2, 1), "l10": (16, 18, 750, 64, 3, 1, 1),
}
def timeit(f, n=reps):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, (Cin, H, W, Cout, k, s, p) in CFG.items():
    if which != "all" and which != name: continue
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    sc = torch.rand(Cin, device="cuda") + 0.5; sh = torch.randn(Cin, device="cuda") * 0.1
    y = ops.conv2d_fwd(x, w, s, p, sc, sh, True)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * k * k
    tf = timeit(lambda: ops.conv2d_fwd(x, w, s, p, sc, sh, True))
    td = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, s, p))
    tw = timeit(lambda: ops.conv2d_wgrad(x, dy, w.shape, s, p, sc, sh, True))
    print("%-4s B=%d  fwd %.3f ms %.1f TF | dgrad %.3f ms %.1f TF | wgrad %.3f ms %.1f TF" % (
        name, B, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9), flush=True)
    if len(sys.argv) > 4:  # also time the no-prologue variants
        tf0 = timeit(lambda: ops.conv2d_fwd(x, w, s, p))
        tw0 = timeit(lambda: ops.conv2d_wgrad(x, dy, w.shape, s, p))
        print("     plain (MODE 0): fwd %.3f ms %.1f TF | wgrad %.3f ms %.1f TF" % (tf0, fl / tf0 / 1e9, tw0, fl / tw0 / 1e9), flush=True)
