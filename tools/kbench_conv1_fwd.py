import torch
from asvspoof2021_air_amd import ops
x = torch.randn(64, 1, 60, 750, device="cuda"); w = torch.randn(16, 1, 9, 3, device="cuda")
for _ in range(8): ops.conv2d_fwd(x, w, (3, 1), (1, 1))
torch.cuda.synchronize()
