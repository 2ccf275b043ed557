"""Split-bf16 stride-2 weight gradient (conv_bf3.hip, option CONV_S2 bit 16) against the direct f32-MFMA kernel."""
import sys
import torch
import torch.nn.functional as F
from asvspoof2021_air_amd import _hip, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CFG = {"l2s": (64, 18, 750, 128), "l3s": (128, 9, 375, 256), "l4s": (256, 5, 188, 512), "odd": (32, 7, 61, 128)}


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, (Cin, H, W, Cout) in CFG.items():
    g = torch.Generator().manual_seed(Cin)
    nb = 3
    xs = torch.relu(torch.randn(nb, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).double().requires_grad_(True)
    y = F.conv2d(xs.double(), w, None, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    ref, = torch.autograd.grad(y, w, dy.double())
    scale = float(ref.abs().max())
    errs = {}
    for label, opt in (("f32", 15), ("bf3", 31)):
        with _hip.options(CONV_S2=opt):
            got = ops.conv2d_wgrad(xs.cuda(), dy.cuda(), tuple(w.shape), 2, 1)
            errs[label] = float((got.cpu().double() - ref).abs().max()) / scale
    Ho, Wo = y.shape[2], y.shape[3]
    x = torch.relu(torch.randn(B, Cin, H, W, device="cuda"))
    dyb = torch.randn(B, Cout, Ho, Wo, device="cuda")
    fl = 2.0 * B * Cout * Ho * Wo * Cin * 9
    t = {}
    for label, opt in (("f32", 15), ("bf3", 31)):
        with _hip.options(CONV_S2=opt):
            t[label] = timeit(lambda: ops.conv2d_wgrad(x, dyb, tuple(w.shape), 2, 1))
    print("%-4s f32 %.3f ms %6.1f TF err %.2e | bf3 %.3f ms %6.1f TF err %.2e | %.2fx" % (
        name, t["f32"], fl / t["f32"] / 1e9, errs["f32"], t["bf3"], fl / t["bf3"] / 1e9, errs["bf3"], t["f32"] / t["bf3"]), flush=True)
