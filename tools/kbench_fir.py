"""Micro-benchmark of the channel-augmentation FIR kernel (B utterances of 4 s, H taps)."""
import sys, torch
from asvspoof2021_air_amd.augment import ir_convolve
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = 64000
for H in (256, 1024, 4096):
    x = 0.1 * torch.randn(B, L, device="cuda")
    irs = torch.randn(30, H, device="cuda") * 0.05
    idx = torch.randint(0, 30, (B,), device="cuda", dtype=torch.int32)
    for norm in (False, True):
        for _ in range(2): ir_convolve(x, irs, idx, norm)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): ir_convolve(x, irs, idx, norm)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("H=%-5d normalize=%d  %.3f ms  %.1f TFLOP/s (fp32 VALU)" % (H, norm, ms, 2.0 * B * L * H / ms / 1e9), flush=True)
