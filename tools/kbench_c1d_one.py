"""One pointwise conv1d pass at a real ECAPA shape, a few launches (for PMC runs).  argv: layer fwd|dgrad [B] [T]"""
import sys, torch
from asvspoof2021_air_amd import ops
CFG = {"c512": (512, 512), "layer4": (1536, 1536), "att0": (1536, 128), "att3": (128, 1536)}
Cin, Cout = CFG[sys.argv[1]]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = int(sys.argv[4]) if len(sys.argv) > 4 else 750
x = torch.randn(B, Cin, T, device="cuda"); w = torch.randn(Cout, Cin, 1, device="cuda") * 0.05
dy = torch.randn(B, Cout, T, device="cuda")
for _ in range(6):
    if sys.argv[2] == "fwd": ops.conv1d_fwd(x, w, relu=True, bf16=True)
    else: ops.conv1d_dgrad(dy, w, bf16=True)
torch.cuda.synchronize()
