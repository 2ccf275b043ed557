"""layer4 forward / dgrad from the bf16 copy (K-major GEMM) vs from the fp32 tensor (transposed conversion + GEMM)."""
import torch
from asvspoof2021_air_amd import ops
B, C, T = 128, 1536, 750
x = torch.randn(B, C, T, device="cuda"); w = torch.randn(C, C, 1, device="cuda") * 0.05; dy = torch.randn(B, C, T, device="cuda")
xb = ops.conv1d_cvt_bf16(x, ops.bf16_rows(None, B, C, T, x.device))
for _ in range(5):
    ops.conv1d_pointwise_kmajor(xb, w, T, relu=True)
    ops.conv1d_pointwise_kmajor(xb, w, T, dgrad=True)
    ops.conv1d_fwd(x, w, relu=True, bf16=True)
    ops.conv1d_dgrad(dy, w, bf16=True)
torch.cuda.synchronize()
