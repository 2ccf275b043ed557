"""Times the bf16-resident dilated K = 3 Res2 convolution (air_h_conv1d_tap) at ECAPA's shape (B = 128, 64 -> 64
channels, T = 750).  AIR_HIP_LIB=<variant .so> selects an A/B build."""
import torch
from asvspoof2021_air_amd import ops, ops_h
B, W, T = 128, 64, 750
x = ops_h.from_f32(torch.randn(B, W, T, device="cuda"))
w = [torch.randn(W, W, 3, device="cuda") * 0.05]
bias = torch.randn(W, device="cuda")
wp = ops.conv1d_tap_pack(w, transpose=False)
out = ops_h.rows(B, W, T, "cuda")
def run(): ops_h.conv_tap(x, wp[0], T, 3, W, W, bias=bias, relu=True, out=out)
for _ in range(50): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): run()
b.record(); torch.cuda.synchronize()
print("conv_tap fwd + bias + relu: %.2f us" % (a.elapsed_time(b) / 200 * 1e3))
