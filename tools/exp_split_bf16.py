"""split-bf16 ("bf16 x 3", six products) for the DIRECT f32 convolutions - evaluation.

(1) accuracy (CPU, numpy): the ResNet's stride-2 / conv5 contractions as GEMMs; fp32 operands split into three bf16 planes
    (hi, mid, lo: 8 + 8 + 8 mantissa bits), the six products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid accumulated in
    fp32, against fp64 - beside the plain fp32 fma chain (what the f32 MFMA computes) and a 3-product variant.
(2) speed proxy (GPU): the library's best bf16 GEMM (c1b_gemm_ps_kernel through ops_h.conv_pointwise) at the im2col shape
    of each layer, x 6 launches, against the direct f32-MFMA kernel of that layer today.  A fused six-product kernel
    shares its operand loads, so 6 x (one bf16 GEMM) is the PESSIMISTIC end; the MFMA-only floor (2500 / 6 = 417 TF) the
    optimistic one.
usage: python tools/exp_split_bf16.py [cpu|gpu|all]"""
import sys

import numpy as np


def bf16(x):
    """round-to-nearest-even to bf16, returned as float32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16(r1)
    lo = bf16((r1 - mid).astype(np.float32))
    return hi, mid, lo


def mm32(a, b, kb=16):
    """fp32 accumulation in K-blocks of 16 (the MFMA's K), block sums added sequentially in fp32"""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], kb):
        acc = (acc + (a[:, k:k + kb].astype(np.float64) @ b[k:k + kb].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc


def chain32(a, b):
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc + a[:, k:k + 1] * b[k:k + 1]).astype(np.float32)  # (product rounded, then the sum: >= the fma's error)
    return acc


def accuracy():
    rng = np.random.default_rng(5)
    print("%-28s %10s %10s %10s %10s" % ("contraction", "f32 chain", "bf16x3(6)", "bf16x3(3)", "bf16x1"))
    for name, K in (("layer2.0.conv1  K = 576", 576), ("layer3.0.conv1  K = 1152", 1152), ("layer4.0.conv1  K = 2304", 2304),
                    ("conv5           K = 4608", 4608), ("1x1 shortcut    K = 256", 256)):
        M, N = 96, 64
        a = rng.standard_normal((M, K)).astype(np.float32)
        a = np.maximum(a, 0) if "conv1" in name or "conv5" in name else a   # activated inputs (post-ReLU)
        b = (rng.standard_normal((K, N)) * np.sqrt(2.0 / K)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        s = np.abs(ref).max()
        ah, am, al = split3(a)
        bh, bm, bl = split3(b)
        six = np.zeros((M, N), np.float32)
        for x, y in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):  # small terms first
            six = (six + mm32(x, y)).astype(np.float32)
        three = np.zeros((M, N), np.float32)
        for x, y in ((am, bh), (ah, bm), (ah, bh)):
            three = (three + mm32(x, y)).astype(np.float32)
        one = mm32(ah, bh)
        e = lambda y: np.abs(y.astype(np.float64) - ref).max() / s
        print("%-28s %10.2e %10.2e %10.2e %10.2e" % (name, e(chain32(a, b)), e(six), e(three), e(one)))


def speed():
    import torch
    from asvspoof2021_air_amd import ops, ops_h

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

    B = 64
    print("%-6s %22s %30s %24s" % ("layer", "direct f32 kernel today", "one bf16 GEMM (im2col shape)", "6 x GEMM vs today"))
    for name, (Cin, H, W, Cout) in {"l2s": (64, 18, 750, 128), "l3s": (128, 9, 375, 256), "l4s": (256, 5, 188, 512)}.items():
        x = torch.randn(B, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        y = ops.conv2d_fwd(x, w, 2, 1)
        fl = 2.0 * y.numel() * Cin * 9
        t_f = timeit(lambda: ops.conv2d_fwd(x, w, 2, 1))
        K, T = Cin * 9, y.shape[2] * y.shape[3]
        xr = ops_h.from_f32(torch.randn(B, K, T, device="cuda"))
        wr = torch.randn(Cout, K, 1, device="cuda") * 0.05
        out = ops_h.rows(B, Cout, T, "cuda")
        t_g = timeit(lambda: ops_h.conv_pointwise(xr, wr, T, out=out))
        print("%-6s %8.3f ms %7.1f TF   %10.3f ms %8.1f TF (K = %4d, Tp = %d)   %8.3f ms = %.2fx" % (
            name, t_f, fl / t_f / 1e9, t_g, fl / t_g / 1e9, K, xr.shape[2], 6 * t_g, t_f / (6 * t_g)), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("cpu", "all"):
        accuracy()
    if what in ("gpu", "all"):
        speed()
