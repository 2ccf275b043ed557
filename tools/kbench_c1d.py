"""Micro-benchmark of the pointwise conv1d layers of ECAPA-TDNN-512 at their real shapes
(B=128, T=750): fp32 kernels vs bf16 kernels, forward / dgrad / wgrad."""
import sys, torch
from asvspoof2021_air_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else "all"
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16"]
T = int(sys.argv[4]) if len(sys.argv) > 4 else 750
CFG = {"c512": (512, 512), "layer4": (1536, 1536), "att0": (1536, 128), "att3": (128, 1536)}
def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, (Cin, Cout) in CFG.items():
    if which != "all" and which != name: continue
    x = torch.randn(B, Cin, T, device="cuda"); w = torch.randn(Cout, Cin, 1, device="cuda") * 0.05
    dy = torch.randn(B, Cout, T, device="cuda")
    fl = 2.0 * B * T * Cin * Cout
    gb = 4.0 * B * T * (Cin + Cout) / 1e9
    for mode in modes:
        bf = mode == "bf16"
        tf = timeit(lambda: ops.conv1d_fwd(x, w, relu=True, bf16=bf))
        td = timeit(lambda: ops.conv1d_dgrad(dy, w, bf16=bf))
        tw = timeit(lambda: ops.conv1d_wgrad(x, dy, w.shape, bf16=bf))
        print("%-6s %-4s fwd %.3f ms %.0f TF %.2f TB/s | dgrad %.3f ms %.0f TF | wgrad %.3f ms %.0f TF %.2f TB/s" % (
            name, mode, tf, fl / tf / 1e9, gb / tf, td, fl / td / 1e9, tw, fl / tw / 1e9, gb / tw), flush=True)
