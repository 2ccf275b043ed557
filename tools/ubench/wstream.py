#!/usr/bin/env python3
"""Write-side HBM micro-benchmarks (measurement tool, not the product path): fill, and row-wise copies of a
(B, C, T) fp32 tensor with T = 750 (rows start on 8-byte boundaries only) against T = 768."""
import torch
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
n = 1 << 28
x = torch.empty(n, device="cuda")
t = timeit(lambda: x.fill_(1.0))
print("fill 1 GiB: %.1f GB/s" % (4 * n / t / 1e9))
for T in (750, 768):
    a = torch.randn(128, 1536, T, device="cuda"); b = torch.empty_like(a)
    t = timeit(lambda: b.fill_(0.5))
    print("fill (128,1536,%d): %.1f GB/s" % (T, a.numel() * 4 / t / 1e9))
    t = timeit(lambda: b.copy_(a))
    print("copy (128,1536,%d): %.1f GB/s r+w" % (T, 2 * a.numel() * 4 / t / 1e9))
    t = timeit(lambda: torch.relu(a, out=b) if False else torch.clamp_min(a, 0.0, out=b))
    print("relu (128,1536,%d): %.1f GB/s r+w" % (T, 2 * a.numel() * 4 / t / 1e9))
