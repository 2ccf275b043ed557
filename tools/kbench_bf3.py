"""Split-bf16 stride-2 forward (conv_bf3.hip, option CONV_S2 bit 4) against the direct f32-MFMA kernel: time at B = 64 and
error against an fp64 CPU convolution (B = 2).  usage: PYTHONPATH=. python tools/kbench_bf3.py [B]"""
import sys
import torch
import torch.nn.functional as F
from asvspoof2021_air_amd import _hip, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CFG = {"l2s": (64, 18, 750, 128), "l3s": (128, 9, 375, 256), "l4s": (256, 5, 188, 512), "odd": (32, 7, 61, 64)}


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
    xs = torch.relu(torch.randn(2, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    ref = F.conv2d(xs.double(), w.double(), None, 2, 1)
    scale = float(ref.abs().max())
    errs = {}
    for label, opt in (("f32", 3), ("bf3", 7)):
        with _hip.options(CONV_S2=opt):
            y = ops.conv2d_fwd(xs.cuda(), w.cuda(), 2, 1)
            errs[label] = float((y.cpu().double() - ref).abs().max()) / scale
    x = torch.relu(torch.randn(B, Cin, H, W, device="cuda"))
    wd = w.cuda()
    fl = 2.0 * B * Cout * ref.shape[2] * ref.shape[3] * Cin * 9
    t = {}
    for label, opt in (("f32", 3), ("bf3", 7)):
        with _hip.options(CONV_S2=opt):
            wp_bytes = int(_hip.lib().air_conv2d_prepack_bytes(__import__("ctypes").byref(ops._conv_desc(x.shape, wd.shape, 2, 1)), 0))
            pk = ops.conv2d_prepack(wd, x.shape, 2, 1, 0)
            t[label] = timeit(lambda: ops.conv2d_fwd(x, wd, 2, 1, w_packed=pk))
    print("%-4s f32 %.3f ms %6.1f TF err %.2e | bf3 %.3f ms %6.1f TF err %.2e | %.2fx" % (
        name, t["f32"], fl / t["f32"] / 1e9, errs["f32"], t["bf3"], fl / t["bf3"] / 1e9, errs["bf3"], t["f32"] / t["bf3"]), flush=True)
