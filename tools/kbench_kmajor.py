"""Pointwise forward / dgrad from the bf16 copy (K-major GEMM) vs from the fp32 tensor, layer4 and the 512-channel layers."""
import sys, torch
from asvspoof2021_air_amd import ops
B, T = 128, 750
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
x = torch.randn(B, C, T, device="cuda"); w = torch.randn(C, C, 1, device="cuda") * 0.05; dy = torch.randn(B, C, T, device="cuda")
xb = ops.conv1d_cvt_bf16(x, ops.bf16_rows(None, B, C, T, x.device))
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print("C=%d  kmajor fwd %.3f ms  dgrad %.3f ms | fp32-input fwd %.3f ms  dgrad %.3f ms" % (
    C, timeit(lambda: ops.conv1d_pointwise_kmajor(xb, w, T, relu=True)), timeit(lambda: ops.conv1d_pointwise_kmajor(xb, w, T, dgrad=True)),
    timeit(lambda: ops.conv1d_fwd(x, w, relu=True, bf16=True)), timeit(lambda: ops.conv1d_dgrad(dy, w, bf16=True))))
