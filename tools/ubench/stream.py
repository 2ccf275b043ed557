#!/usr/bin/env python3
"""HBM stream micro-benchmarks on the GPU box (confirm the nominal peaks the rooflines divide by).
copy: y = x; triad: a = b + s * c, over 1 GiB fp32 tensors (far beyond the 256 MiB Infinity Cache), plus rocminfo's
CU count and clocks.  Uses plain torch ops: this is a measurement tool, not the product path."""
import subprocess, torch
n = 1 << 28  # 1 GiB of fp32
x = torch.empty(n, device="cuda").normal_(); y = torch.empty_like(x); z = torch.empty_like(x)
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
t = timeit(lambda: y.copy_(x))
print("copy : %.1f GB/s (read + write)" % (2 * 4 * n / t / 1e9))
t = timeit(lambda: torch.add(y, z, alpha=0.5, out=x))
print("triad: %.1f GB/s (2 reads + 1 write)" % (3 * 4 * n / t / 1e9))
t = timeit(lambda: x.sum())
print("read : %.1f GB/s (reduction)" % (4 * n / t / 1e9))
try:
    out = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=60).stdout
    for ln in out.splitlines():
        if any(k in ln for k in ("Marketing Name", "Compute Unit", "Max Clock Freq", "gfx9")) and "CPU" not in ln:
            print("rocminfo:", ln.strip())
except Exception as exc:  # noqa
    print("rocminfo unavailable:", exc)
