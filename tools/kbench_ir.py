"""Times the impulse-response convolution (air_ir_convolve) at configs[4]'s shape: overlap-save FFT against the direct FIR."""
import torch
from asvspoof2021_air_amd import _hip
from asvspoof2021_air_amd.augment import ir_convolve, synthetic_ir_bank


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


B, L = 128, 64000
x = 0.1 * torch.randn(B, L, device="cuda")
irs = synthetic_ir_bank().cuda()
idx = (torch.arange(B, device="cuda", dtype=torch.int32) % 30)
for mode in (1, 0):
    _hip.set_option("IR_FFT", mode)
    y = ir_convolve(x, irs, idx, True)
    print("IR_FFT=%d: %.1f us per call (B = %d, L = %d, H = %d)" % (mode, t(lambda: ir_convolve(x, irs, idx, True)), B, L, irs.shape[1]))
    if mode == 1:
        y1 = y.clone()
    else:
        print("max |fft - direct| = %.3e of peak %.3f" % (float((y1 - y).abs().max()), float(y.abs().max())))
