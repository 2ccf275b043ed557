"""Stride-2 3x3 data gradient (+ the 1x1 shortcut's) at the ResNet's three downsampling blocks: the one-pass kernel
(option CONV_S2 bit 2) against the four class launches + the shortcut's read-modify-write launch (rounds 1-3).
usage: python tools/kbench_s2_dgrad.py [B] [reps]"""
import sys, torch
from asvspoof2021_air_amd import ops, _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CFG = {"l2": (64, 18, 750, 128), "l3": (128, 9, 375, 256), "l4": (256, 5, 188, 512)}
def timeit(f, n=reps):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, (Cin, H, W, Cout) in CFG.items():
    xs = (B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05; wsc = torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.1
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(B, Cout, Ho, Wo, device="cuda"); dysc = torch.randn_like(dy)
    fl = 2.0 * dy.numel() * Cin * 9
    def old():
        d = ops.conv2d_dgrad(dy, w, xs, 2, 1)
        return ops.conv2d_dgrad(dysc, wsc, xs, 2, 0, accumulate=d, out=d)
    with _hip.options(CONV_S2=1):
        r0 = ops.conv2d_dgrad(dy, w, xs, 2, 1).clone(); r1 = old().clone()
        t0 = timeit(lambda: ops.conv2d_dgrad(dy, w, xs, 2, 1)); t1 = timeit(old)
    n0 = ops.conv2d_dgrad(dy, w, xs, 2, 1); n1 = ops.conv2d_dgrad_s2_pair(dy, w, dysc, wsc, xs)
    pk = ops.conv2d_dgrad_s2_pair_prepack(w, wsc, xs)
    n2 = ops.conv2d_dgrad_s2_pair(dy, w, dysc, wsc, xs, packed=pk)
    e0 = ((n0 - r0).abs().max() / r0.abs().max()).item(); e1 = ((n1 - r1).abs().max() / r1.abs().max()).item()
    e2 = (n2 - n1).abs().max().item()
    u0 = timeit(lambda: ops.conv2d_dgrad(dy, w, xs, 2, 1)); u1 = timeit(lambda: ops.conv2d_dgrad_s2_pair(dy, w, dysc, wsc, xs))
    u2 = timeit(lambda: ops.conv2d_dgrad_s2_pair(dy, w, dysc, wsc, xs, packed=pk))
    print("%s 3x3: classes %.3f ms %5.1f TF | one pass %.3f ms %5.1f TF (d %.1e) || + shortcut: %.3f ms | pair %.3f ms, "
          "prepacked %.3f ms %5.1f TF (d %.1e, %.1e)" % (name, t0, fl / t0 / 1e9, u0, fl / u0 / 1e9, e0, t1, u1, u2,
                                                         fl * 10 / 9 / u2 / 1e9, e1, e2), flush=True)
