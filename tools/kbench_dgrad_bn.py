"""Micro-benchmark: the 3x3 / stride 1 data gradient with and without the BatchNorm-backward sums in its epilogue
(air_conv2d_dgrad_bn), and the BatchNorm backward with and without them, at the ResNet's layer shapes, B = 64."""
import sys, torch
from asvspoof2021_air_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CFG = {"l1": (64, 18, 750), "l2": (128, 9, 375), "l3": (256, 5, 188), "l4": (512, 3, 94)}
def timeit(f, n=reps):
    for _ in range(n): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, (C, H, W) in CFG.items():
    B = 64
    x = torch.randn(B, C, H, W, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
    dy = torch.randn(B, C, H, W, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, invstd, _, _ = ops.bn_stats(x, g, b)
    bn = (x, mean, invstd, g, b)
    t0 = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1))
    t1 = timeit(lambda: ops.conv2d_dgrad(dy, w, x.shape, 1, 1, bn=bn))
    dA, sums = ops.conv2d_dgrad(dy, w, x.shape, 1, 1, bn=bn)
    t2 = timeit(lambda: ops.bn_bwd(x, dA, mean, invstd, g, b, relu=True))
    t3 = timeit(lambda: ops.bn_bwd(x, dA, mean, invstd, g, b, relu=True, sums_in=sums))
    print("%s dgrad %.0f us, with sums %.0f us (+%.0f) | bn_bwd %.0f us, with sums_in %.0f us (-%.0f)" % (
        name, t0, t1, t1 - t0, t2, t3, t2 - t3), flush=True)
