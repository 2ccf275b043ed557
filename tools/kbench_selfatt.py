import torch
from asvspoof2021_air_amd import ops
B, C, T = 64, 256, 94
x = torch.randn(B, C, T, device="cuda"); att = torch.randn(C, device="cuda"); noise = 1e-5 * torch.randn(B, T, C, device="cuda")
for _ in range(6):
    out, alpha = ops.selfatt_pool_fwd(x, att, noise)
    ops.selfatt_pool_bwd(x, att, noise, alpha, out, torch.randn_like(out))
torch.cuda.synchronize()
