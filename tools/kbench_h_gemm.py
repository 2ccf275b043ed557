"""Times the bf16-resident pointwise GEMM (air_h_conv1d_pointwise / air_h_conv1d_wgrad) at ECAPA's shapes.
AIR_HIP_LIB=<variant .so> selects an A/B build (asvspoof2021_air_amd/build.py --variant)."""
import sys
import torch
from asvspoof2021_air_amd import ops_h


def timeit(fn, reps=30):
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
    rows.append(("fwd bias+relu", timeit(lambda: ops_h.conv_pointwise(x, w, T, bias=bias, relu=True, out=out)), fl))
    rows.append(("fwd plain", timeit(lambda: ops_h.conv_pointwise(x, w, T, out=out)), fl))
    rows.append(("dgrad + acc", timeit(lambda: ops_h.conv_pointwise(x, w, T, dgrad=True, acc=r1, out=out)), fl))
    rows.append(("dgrad + acc + acc2", timeit(lambda: ops_h.conv_pointwise(x, w, T, dgrad=True, acc=r1, acc2=r2, out=out)), fl))
    rows.append(("wgrad 512x512", timeit(lambda: ops_h.conv_wgrad(x, r1, T, dw)), fl))
    x4 = ops_h.from_f32(torch.randn(B, C4, T, device=dev, generator=g))
    w4 = torch.randn(C4, C4, 1, device=dev, generator=g) * 0.05
    o4 = torch.empty_like(x4)
    rows.append(("layer4 fwd 1536x1536", timeit(lambda: ops_h.conv_pointwise(x4, w4, T, relu=True, out=o4), 10), 2.0 * B * T * C4 * C4))
    for name, us, f in rows:
        print("%-24s %8.1f us  %7.1f TF" % (name, us, f / us / 1e6))


if __name__ == "__main__":
    sys.exit(main())
