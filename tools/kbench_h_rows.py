"""Bandwidth of the bf16-resident row kernels (one wave per row, 8-byte accesses) against torch.Tensor.copy_ at ECAPA's three tensor sizes.
usage (GPU box): python tools/kbench_h_rows.py"""
import torch
from asvspoof2021_air_amd import ops_h as oh
def timeit(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B, T = 128, 750
for C in (1536, 512, 64):
    x = oh.rows(B, C, T, "cuda", zero=True); y = oh.rows(B, C, T, "cuda", zero=True)
    nb = x.numel() * 2
    t = timeit(lambda: oh.copy(x, y))
    t2 = timeit(lambda: y.copy_(x))
    sc = torch.ones(C, device="cuda"); sh = torch.zeros(C, device="cuda")
    t3 = timeit(lambda: oh.bn_apply(x, T, sc, sh, out=y))
    print("C=%d  %.1f MB: h_copy %.1f us %.2f TB/s | torch copy_ %.1f us %.2f TB/s | h_bn_apply %.1f us %.2f TB/s" % (C, nb / 1e6, t, 2 * nb / t / 1e6, t2, 2 * nb / t2 / 1e6, t3, 2 * nb / t3 / 1e6))
