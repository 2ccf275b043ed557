"""One shape of the bf16-resident pointwise GEMM, launched 20 times (for tools/pmc_kernel.sh)."""
import torch
from asvspoof2021_air_amd import ops_h
B, C, T = 128, 512, 750
g = torch.Generator(device="cuda").manual_seed(1)
x = ops_h.from_f32(torch.randn(B, C, T, device="cuda", generator=g))
w = torch.randn(C, C, 1, device="cuda", generator=g) * 0.05
out = torch.empty_like(x)
for _ in range(20):
    ops_h.conv_pointwise(x, w, T, out=out)
torch.cuda.synchronize()
