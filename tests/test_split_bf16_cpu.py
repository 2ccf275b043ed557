"""The arithmetic claim behind csrc/conv_bf3.hip, checked without a GPU (numpy emulation, tools/exp_split_bf16.py):
(1) a finite fp32 value splits EXACTLY into three bf16 values, x = hi + mid + lo (round to nearest even at each step);
(2) the six-product form hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid with fp32 accumulation per 16-term block is no
    farther from fp64 than the fp32 multiply-add chain it replaces, on the ResNet's stride-2 contraction lengths, and an
    order of magnitude inside the direct kernels' parity bound (tests/_budget.py: conv_rtol 1e-5);
(3) the three-product form is NOT fp32-equivalent (it would be narrower than the reference)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import exp_split_bf16 as E  # noqa: E402

from _budget import STRICT  # noqa: E402


def test_split_is_exact():
    """Exact on 2^-100 <= |x| <= 3.38e38 and 0.  Outside (stated in conv_bf3.hip): a value within 2^-8 of FLT_MAX rounds its
    hi plane to bf16 infinity, and the low planes of values below ~2^-110 fall into bf16's denormal range - neither occurs
    in activations or weights (BatchNorm keeps them O(1)), and the fp32 products of the latter underflow anyway."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, 200000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 3.38e38, -3.38e38, 2.0 ** -100, -2.0 ** -100, 0.1, 1 / 3, 65504.0,
                                  1.0 - 2.0 ** -24], np.float32)])
    hi, mid, lo = E.split3(x)
    for p in (hi, mid, lo):  # each piece IS a bf16 (its low 16 bits are zero)
        assert not np.any(p.view(np.uint32) & 0xFFFF)
    # exactly, not just after rounding back to fp32
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    # the stated edge: FLT_MAX's hi plane overflows
    with np.errstate(invalid="ignore", over="ignore"):
        h, _, _ = E.split3(np.array([3.4e38], np.float32))
    assert np.isinf(h[0])


def test_six_products_are_fp32_equivalent_three_are_not():
    rng = np.random.default_rng(5)
    for K in (576, 1152, 2304):
        M, N = 64, 48
        a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
        b = (rng.standard_normal((K, N)) * np.sqrt(2.0 / K)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        s = np.abs(ref).max()
        ah, am, al = E.split3(a)
        bh, bm, bl = E.split3(b)
        six = np.zeros((M, N), np.float32)
        for x, y in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            six = (six + E.mm32(x, y)).astype(np.float32)
        three = np.zeros((M, N), np.float32)
        for x, y in ((am, bh), (ah, bm), (ah, bh)):
            three = (three + E.mm32(x, y)).astype(np.float32)
        e6 = np.abs(six - ref).max() / s
        e3 = np.abs(three - ref).max() / s
        e32 = np.abs(E.chain32(a, b) - ref).max() / s
        assert e6 <= 1.5 * e32 and e6 <= 0.1 * STRICT["conv_rtol"], (K, e6, e32)
        assert e3 >= 3.0 * e32, (K, e3, e32)  # the narrower form shows
