"""Times the bf16-resident pointwise GEMM (air_h_conv1d_pointwise / air_h_conv1d_wgrad) at ECAPA's shapes.
AIR_HIP_LIB=<variant .so> selects an A/B build (asvspoof2021_air_amd/build.py --variant)."""
import os
import sys
import torch
from asvspoof2021_air_amd import ops_h


def timeit(fn, reps=100):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    B, C, T = 128, 512, 750
    C4 = 1536
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops_h.from_f32(torch.randn(B, C, T, device=dev, generator=g))
    r1 = ops_h.from_f32(torch.randn(B, C, T, device=dev, generator=g))
    r2 = ops_h.from_f32(torch.randn(B, C, T, device=dev, generator=g))
    w = torch.randn(C, C, 1, device=dev, generator=g) * 0.05
    bias = torch.randn(C, device=dev, generator=g)
    out = torch.empty_like(x)
    dw = torch.empty_like(w)
    fl = 2.0 * B * T * C * C
    rows = []
    only = os.environ.get("KB_ONLY", "")  # "layer4": just the K = 1536 launch (PMC passes)
    if only != "layer4":
      rows.append(("fwd bias+relu", timeit(lambda: ops_h.conv_pointwise(x, w, T, bias=bias, relu=True, out=out)), fl))
      rows.append(("fwd plain", timeit(lambda: ops_h.conv_pointwise(x, w, T, out=out)), fl))
      rows.append(("dgrad + acc", timeit(lambda: ops_h.conv_pointwise(x, w, T, dgrad=True, acc=r1, out=out)), fl))
      rows.append(("dgrad + acc + acc2", timeit(lambda: ops_h.conv_pointwise(x, w, T, dgrad=True, acc=r1, acc2=r2, out=out)), fl))
      rows.append(("wgrad 512x512", timeit(lambda: ops_h.conv_wgrad(x, r1, T, dw)), fl))
    x4 = ops_h.from_f32(torch.randn(B, C4, T, device=dev, generator=g))
    w4 = torch.randn(C4, C4, 1, device=dev, generator=g) * 0.05
    o4 = torch.empty_like(x4)
    if os.environ.get("KB_TRACE"):  # a -DG2_X_TRACE build: per-stage s_memtime deltas of workgroup 0, waves 0 and 4
        import numpy as np
        tb = torch.zeros(2 * (4096 + 256 * 2 * 4) + 64, device=dev)
        for _ in range(20):
            ops_h.conv_pointwise(x4, w4, T, bias=tb, relu=True, out=o4)
        torch.cuda.synchronize()
        raw = tb.cpu().numpy().view(np.uint64)
        for wv in range(2):
            t = raw[1024 + wv * 512: 1024 + wv * 512 + 512].reshape(128, 4).astype(np.int64)
            print("wave %d: stage | wait+barrier | dma issue | mfma+reads | stage total (core clocks)" % (4 * wv))
            for i in range(0, 60):
                nxt = t[i + 1][0] if i + 1 < 128 else t[i][3]
                print("  %3d  %6d %6d %6d   %6d" % (i, t[i][1] - t[i][0], t[i][2] - t[i][1], t[i][3] - t[i][2], nxt - t[i][0]))
        tot = raw[4096: 4096 + 256 * 8].reshape(256, 2, 4).astype(np.int64)
        print("per workgroup totals (core clocks): wait+barrier | dma issue | mfma+reads | kernel, waves 0 / 4")
        for wv in range(2):
            q = tot[:, wv, :]
            print("  wave %d: mean %s  min %s  max %s" % (4 * wv, q.mean(0).astype(int), q.min(0), q.max(0)))
        order = np.argsort(tot[:, 0, 3])
        for b in list(order[:4]) + list(order[-4:]):
            print("  wg %3d (xcd %d): %s | %s" % (b, b % 8, tot[b, 0], tot[b, 1]))
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops_h.conv_pointwise(x4, w4, T, bias=tb, relu=True, out=o4)
        b2.record()
        torch.cuda.synchronize()
        print("kernel %.1f us" % (a.elapsed_time(b2) / 20 * 1e3))
        return
    rows.append(("layer4 fwd 1536x1536", timeit(lambda: ops_h.conv_pointwise(x4, w4, T, relu=True, out=o4), 60), 2.0 * B * T * C4 * C4))
    for name, us, f in rows:
        print("%-24s %8.1f us  %7.1f TF" % (name, us, f / us / 1e6))


if __name__ == "__main__":
    sys.exit(main())
