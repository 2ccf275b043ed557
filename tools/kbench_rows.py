"""Micro-benchmark of the wave-per-row ECAPA kernels at their real shapes (B = 128, T = 750)."""
import torch
from asvspoof2021_air_amd import ops
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
B, T = 128, 750
x = torch.randn(B, 512, T, device="cuda"); z = torch.randn(B, 512, device="cuda"); res = torch.randn_like(x); out = torch.empty_like(x)
t = timeit(lambda: ops.se_scale_fwd(x, z, res, out)); print("se_scale_fwd  %.3f ms %.2f TB/s" % (t, 3 * x.numel() * 4 / t / 1e9))
t = timeit(lambda: ops.se_scale_bwd(x, z, res)); print("se_scale_bwd  %.3f ms %.2f TB/s" % (t, 3 * x.numel() * 4 / t / 1e9))
x4 = torch.randn(B, 1536, T, device="cuda").relu_(); n4 = x4.numel() * 4
t = timeit(lambda: ops.row_stats(x4, True, 1e-4)); print("row_stats     %.3f ms %.2f TB/s" % (t, n4 / t / 1e9))
mean, std = ops.row_stats(x4, True, 1e-4); dm = torch.randn_like(mean); ds = torch.randn_like(std); dx = torch.randn_like(x4)
rows = torch.empty_like(mean)
t = timeit(lambda: ops.row_stats_bwd(x4, mean, std, dm, ds, dx, accumulate=True, relu_mask=True, rowsum=rows)); print("row_stats_bwd %.3f ms %.2f TB/s" % (t, 3 * n4 / t / 1e9))
lg = torch.randn_like(x4)
t = timeit(lambda: ops.asp_fwd(x4, lg.clone())); t0 = timeit(lambda: lg.clone()); print("asp_fwd       %.3f ms %.2f TB/s (clone %.3f excluded)" % (t - t0, 3 * n4 / (t - t0) / 1e9, t0))
w = lg.clone(); pooled = ops.asp_fwd(x4, w); dp = torch.randn_like(pooled)
t = timeit(lambda: ops.asp_bwd(x4, w.clone(), pooled, dp, dx, accumulate=False, rowsum=rows)); print("asp_bwd       %.3f ms %.2f TB/s (clone excluded)" % (t - t0, 4 * n4 / (t - t0) / 1e9))
