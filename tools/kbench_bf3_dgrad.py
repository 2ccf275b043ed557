"""Split-bf16 paired stride-2 data gradient (conv_bf3.hip, option CONV_S2 bit 8) against the one-pass f32 kernel."""
import sys
import torch
import torch.nn.functional as F
from asvspoof2021_air_amd import _hip, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CFG = {"l2s": (64, 18, 750, 128), "l3s": (128, 9, 375, 256), "l4s": (256, 5, 188, 512), "odd": (64, 7, 61, 64)}


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
    xs = torch.randn(2, Cin, H, W, generator=g).double().requires_grad_(True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    wsc = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
    y = F.conv2d(xs, w.double(), None, 2, 1)
    ysc = F.conv2d(xs, wsc.double(), None, 2, 0)
    dy, dysc = torch.randn(y.shape, generator=g), torch.randn(y.shape, generator=g)
    ref, = torch.autograd.grad([y, ysc], xs, [dy.double(), dysc.double()])
    scale = float(ref.abs().max())
    errs = {}
    for label, opt in (("f32", 3), ("bf3", 15)):
        with _hip.options(CONV_S2=opt):
            got = ops.conv2d_dgrad_s2_pair(dy.cuda(), w.cuda(), dysc.cuda(), wsc.cuda(), (2, Cin, H, W))
            errs[label] = float((got.cpu().double() - ref).abs().max()) / scale
    Ho, Wo = y.shape[2], y.shape[3]
    dyb, dyscb = torch.randn(B, Cout, Ho, Wo, device="cuda"), torch.randn(B, Cout, Ho, Wo, device="cuda")
    fl = 2.0 * B * Cout * Ho * Wo * Cin * 10
    t = {}
    for label, opt in (("f32", 3), ("bf3", 15)):
        with _hip.options(CONV_S2=opt):
            pk = ops.conv2d_dgrad_s2_pair_prepack(w.cuda(), wsc.cuda(), (B, Cin, H, W))
            t[label] = timeit(lambda: ops.conv2d_dgrad_s2_pair(dyb, w.cuda(), dyscb, wsc.cuda(), (B, Cin, H, W), packed=pk))
    print("%-4s f32 %.3f ms %6.1f TF err %.2e | bf3 %.3f ms %6.1f TF err %.2e | %.2fx" % (
        name, t["f32"], fl / t["f32"] / 1e9, errs["f32"], t["bf3"], fl / t["bf3"] / 1e9, errs["bf3"], t["f32"] / t["bf3"]), flush=True)
