"""Times the bf16-row BatchNorm kernels (air_h_bn_stats / apply / bwd) at ECAPA's two shapes."""
import sys

import torch
from asvspoof2021_air_amd import ops_h as oh


def timeit(f, n=50):
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


SHAPES = ((128, 64, 750), (128, 512, 750))
if len(sys.argv) > 1:  # one shape only (per-shape kernel times under tools/kstat.sh)
    SHAPES = (SHAPES[int(sys.argv[1])],)
for B, C, T in SHAPES:
    x = oh.from_f32(torch.randn(B, C, T, device="cuda").relu_())
    dy = oh.from_f32(torch.randn(B, C, T, device="cuda"))
    dy2 = oh.from_f32(torch.randn(B, C, T, device="cuda"))
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    st = oh.bn_stats(x, T, g, b)
    dg, db, dbias = (torch.empty(C, device="cuda") for _ in range(3))
    dx = torch.empty_like(x)
    mb = x.numel() * 2 / 1e6
    ts = timeit(lambda: oh.bn_stats(x, T, g, b))
    ta = timeit(lambda: oh.bn_apply(x, T, st[2], st[3], out=dx))
    tb = timeit(lambda: oh.bn_bwd(x, dy, T, st[0], st[1], g, dg, db, dx=dx, dy2=dy2, dbias=dbias))
    print("(%d, %d, %d) %.1f MB per tensor: stats (partial + finalize) %.1f us, apply %.1f us, bwd (partial + finalize + apply, "
          "dy + dy2, bias gradient) %.1f us" % (B, C, T, mb, ts, ta, tb))
