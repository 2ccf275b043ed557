"""Test infrastructure: where the tolerances of the Winograd (default) configuration come from.

The library has two configurations of the fp32 ResNet convolutions:

* ``strict``: dispatch option NO_WINO4 = 1 - round 1's configuration: 3x3 stride-1 layers on Winograd F(2x2,3x3)
  (rounds BELOW the direct kernel: 3e-7 .. 7e-7 of the output scale), everything else on the direct f32-MFMA
  kernels (fmaf chains: summation order only).  It has to pass the ROUND-1 constants (``STRICT``).
  ``direct`` (NO_WINOGRAD = 3, no Winograd kernel at all) is exercised at kernel level.
* ``default``: Winograd F(4x4,3x3) / F(3x4,3x3) forward + dgrad and F(3x3,2x2) weight gradients.  Their transforms
  round at a larger multiple of the output scale.  The allowance is not a hand-set constant: it is the strict
  constant times the ratio of the two kernels' per-convolution rounding errors, computed here by a numpy
  emulation (fp32 transforms, sequential fp32 accumulation over the input channels, against fp64) on the
  ResNet's four layer shapes, root-sum-squared over the layers, times a margin of 1.1 (kernel-level bound: worst layer x 1.5).  The kernel-level tests
  hold the GPU kernels to the same emulated budget, so neither a regression of the direct kernels (strict run,
  fixed constants) nor of the Winograd kernels (kernel-level bound) can hide inside the model-level allowance.

Matrices: the textbook points (0, +-1, +-2, inf) of conv_wino4.hip for 4 outputs, (0, +-1, 2, inf) for 3.
"""
import contextlib
import functools
import json
import os

import numpy as np

STRICT = {
    "step2_rtol": 1e-4,        # tests/test_resnet_gpu.py::test_trajectory_vs_golden, loss of step 2
    "g_center_atol": 5e-6,     # ::test_grads_vs_oracle_small, OC-Softmax centre gradient: the bound itself (rtol 0)
                               # on entries up to 1.75 (round 1 wrote 1e-6 next to an rtol of 1e-3 that did the
                               # binding; measured 3.2e-6 strict, 1.1e-5 default)
    "full_size_slack": 1e-3,   # tests/test_full_size_gpu.py: e_hip <= 3 e_cpu + slack
    "adv_rel_max": 2e-3,       # tests/test_adversarial.py: conv1 / fc gradient, relative to max
    "conv_rtol": 1e-5,         # tests/test_kernels_gpu.py: one 3x3 convolution, of the output scale
    "running_stat_atol": 1e-5,  # tests/test_resnet_gpu.py::test_forward_vs_golden, BatchNorm running statistics (~1.5)
    "grad_max_entry": 5e-2,    # ::test_grads_vs_oracle_small, worst entry of a gradient tensor relative to its largest
    "grad_rel_l2": 5e-3,       # ::test_grads_vs_oracle_small, per-tensor relative L2 of the gradients at B = 2 (ReLU
                               # sign flips of pre-activations within rounding of 0: their number grows with the rounding)
}
MARGIN = 1.5        # kernel-level bound over the emulated worst layer
MODEL_MARGIN = 1.1  # model-level allowance over strict x ratio

G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
               [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
# F(3,3), points (0, 1, -1, 2, inf)
G3 = np.array([[1 / 2, 0, 0], [-1 / 2, -1 / 2, -1 / 2], [-1 / 6, 1 / 6, -1 / 6], [1 / 6, 1 / 3, 2 / 3], [0, 0, 1]])
BT3 = np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]],
               dtype=np.float64)
AT3 = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]], dtype=np.float64)


def _selfcheck(AT, G, BT):
    rng = np.random.default_rng(0)
    m, n = AT.shape
    d, g = rng.standard_normal(n), rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
    return np.abs(y - ref).max()


assert _selfcheck(AT4, G4, BT4) < 1e-12 and _selfcheck(AT3, G3, BT3) < 1e-12


def conv_ref64(x, w):
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2))
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], H, W))
    for a in range(3):
        for b in range(3):
            y += np.einsum("mc,chw->mhw", w[:, :, a, b], xp[:, a:a + H, b:b + W])
    return y


def conv_direct32(x, w):
    """fp32 chain over (ci, tap), the order of the direct MFMA kernels up to tap order."""
    f = np.float32
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2), f)
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((w.shape[0], H, W), f)
    w = w.astype(f)
    for c in range(C):
        for a in range(3):
            for b in range(3):
                y = (y + w[:, c, a, b][:, None, None] * xp[c, a:a + H, b:b + W][None]).astype(f)
    return y


def conv_wino32(x, w, mh=4, mw=4):
    """F(mh x mw, 3x3) with fp32 transforms and a sequential fp32 sum over the input channels."""
    f = np.float32
    mats = {4: (AT4, G4, BT4), 3: (AT3, G3, BT3)}
    ATh, Gh, BTh = (m.astype(f) for m in mats[mh])
    ATw, Gw, BTw = (m.astype(f) for m in mats[mw])
    C, H, W = x.shape
    M = w.shape[0]
    nh, nw = mh + 2, mw + 2
    TH, TW = -(-H // mh), -(-W // mw)
    xp = np.zeros((C, TH * mh + 2, TW * mw + 2), f)
    xp[:, 1:H + 1, 1:W + 1] = x.astype(f)
    # the kernel's weight transform runs in double and rounds once
    U = np.einsum("ia,mcab,jb->mcij", mats[mh][1], w.astype(np.float64), mats[mw][1]).astype(f)
    V = np.zeros((C, TH, TW, nh, nw), f)
    for i in range(TH):
        for j in range(TW):
            d = xp[:, i * mh:i * mh + nh, j * mw:j * mw + nw]
            t = np.einsum("cab,jb->caj", d, BTw).astype(f)       # along the row first, as the kernel does
            V[:, i, j] = np.einsum("ia,caj->cij", BTh, t).astype(f)
    Mm = np.zeros((M, TH, TW, nh, nw), f)
    for c in range(C):
        Mm = (Mm + U[:, c][:, None, None] * V[c][None]).astype(f)
    t = np.einsum("ia,mhwab->mhwib", ATh, Mm).astype(f)
    Y = np.einsum("mhwib,jb->mhwij", t, ATw).astype(f)
    return Y.transpose(0, 1, 3, 2, 4).reshape(M, TH * mh, TW * mw)[:, :H, :W]


# (Cin, H, W-sample, Cout-sample, tile rows the library picks for this H): layer1 .. layer4 of the ResNet
LAYERS = [(64, 18, 48, 16, 4), (128, 9, 48, 16, 3), (256, 5, 48, 16, 3), (512, 3, 48, 16, 3)]


@functools.lru_cache(maxsize=None)
def conv_budget():
    """Per-layer emulated rounding error / output scale of the direct and the Winograd convolution."""
    rng = np.random.default_rng(1)
    out = []
    for C, H, W, M, mh in LAYERS:
        x = rng.standard_normal((C, H, W))
        w = rng.standard_normal((M, C, 3, 3)) * np.sqrt(2.0 / (9 * C))
        ref = conv_ref64(x, w)
        s = np.abs(ref).max()
        out.append({"Cin": C, "H": H,
                    "direct": float(np.abs(conv_direct32(x, w) - ref).max() / s),
                    "wino44": float(np.abs(conv_wino32(x, w, 4, 4) - ref).max() / s),
                    "wino34": float(np.abs(conv_wino32(x, w, 3, 4) - ref).max() / s)})
    return out


@functools.lru_cache(maxsize=None)
def winograd_ratio():
    """Root-sum-square over the four layers of the Winograd error, over the same for the direct kernel."""
    b = conv_budget()
    w = np.sqrt(sum(max(r["wino44"], r["wino34"]) ** 2 for r in b))
    d = np.sqrt(sum(r["direct"] ** 2 for r in b))
    return max(1.0, float(w / d))


def wino_conv_bound():
    """Kernel-level bound for one Winograd convolution: the worst emulated layer error times the margin."""
    return MARGIN * max(max(r["wino44"], r["wino34"]) for r in conv_budget())


# ---- the ReLU-flip budget (VERDICT r4 item 9b / ADVICE r4): WHICH pre-activations may take the other branch of a ReLU
# is tied to the emulated rounding of the convolutions in front of it, per ReLU, instead of one flat constant.
FLIP_K = 1.0        # a flipped pre-activation lies within FLIP_K x (accumulated emulated conv error x the ReLU input's scale)
                    # of 0.  Measured at BASELINE's full size (64 x 4 s): the worst of the 18 ReLUs reaches 0.33 of it.
FLIP_COUNT_K = 0.25  # flips of a ReLU <= FLIP_COUNT_K x units x 0.8 x that error + FLIP_COUNT_FLOOR: 0.8 d = P(|N(0,1)| < d),
                    # and the emulated error is a MAXIMUM over a layer - the mean |error| of a unit is about a fifth of
                    # it.  Measured: 0.01 - 0.18 of units x 0.8 x error (round 4's flat constants: 2500 flips / 2e-4).
FLIP_COUNT_FLOOR = 4


@functools.lru_cache(maxsize=None)
def relu_flip_eps(path="default"):
    """Per ReLU of the ResNet-18 in execution order (stem, (bn1, bn2) of the 8 blocks, bn5): the emulated rounding error
    of the convolutions in front of it, relative to their output scale, accumulated root-sum-square along the network
    (BatchNorm renormalises every layer's output, so relative errors carry over at O(1) gain).  3x3 stride-1 layers:
    the Winograd error of the layer under ``default`` (tile rows as the library picks them), the direct kernel's under
    ``strict``; stem, stride-2 and 1x1 convolutions: the direct kernel's."""
    b = conv_budget()
    layer_of_block = [0, 0, 1, 1, 2, 2, 3, 3]
    acc2 = b[0]["direct"] ** 2  # the stem convolution
    out = [acc2 ** 0.5]
    for blk, li in enumerate(layer_of_block):
        r, mh = b[li], LAYERS[li][4]
        wino = r["wino44" if mh == 4 else "wino34"] if path == "default" else r["direct"]
        first_of_strided = blk % 2 == 0 and li > 0
        out.append(acc2 ** 0.5)                                   # bn1: the block's input
        acc2 += (r["direct"] if first_of_strided else wino) ** 2  # conv1
        out.append(acc2 ** 0.5)                                   # bn2: conv1's output
        acc2 += wino ** 2 + (r["direct"] ** 2 if first_of_strided or blk == 0 else 0.0)  # conv2 (+ the 1x1 shortcut)
    acc2 += b[3]["direct"] ** 2  # conv5
    out.append(acc2 ** 0.5)
    return out


def check_relu_flips(probe, path, label=""):
    """Every ReLU of the run under test: a unit whose decision differs from the fp64 oracle's had a pre-activation
    within FLIP_K x eps_i x scale_i of 0, and there are at most FLIP_COUNT_K x units_i x 0.8 x eps_i x scale_i (+ floor)
    of them - eps_i from relu_flip_eps(), scale_i = the largest |pre-activation| of that ReLU in the oracle's run.
    Returns the per-ReLU table (also printed) for the record."""
    eps = relu_flip_eps("default" if path == "default" else "strict")
    assert len(probe.flips) == len(eps) == 18
    rows = []
    for i, (nf, mag, sc, n) in enumerate(zip(probe.flips, probe.flip_mag, probe.scale, probe.count)):
        mag_bound = FLIP_K * eps[i] * sc
        cnt_bound = FLIP_COUNT_K * n * 0.8 * eps[i] * sc + FLIP_COUNT_FLOOR
        rows.append({"relu": i, "flips": nf, "units": n, "max_abs_preact": mag, "scale": sc, "eps": eps[i],
                     "mag_bound": mag_bound, "count_bound": cnt_bound})
    print("ReLU-flip budget %s[%s]: relu flips/bound  |preact|/bound" % (label, path))
    for r in rows:
        print("  %2d  %6d / %8.1f   %.2e / %.2e" % (r["relu"], r["flips"], r["count_bound"], r["max_abs_preact"], r["mag_bound"]))
    for r in rows:
        assert r["max_abs_preact"] <= r["mag_bound"], r
        assert r["flips"] <= r["count_bound"], r
    return rows


def tol(key, path):
    """Tolerance of a model-level assertion under ``path`` = 'strict' | 'default'."""
    if path == "strict":
        return STRICT[key]
    return STRICT[key] * winograd_ratio() * MODEL_MARGIN


@contextlib.contextmanager
def conv_path(path):
    """Run a block under the strict (all-direct) or the default (Winograd) convolution configuration."""
    from asvspoof2021_air_amd import _hip
    if path == "strict":
        with _hip.options(NO_WINO4=1, CONV_S2=0):  # (CONV_S2 = 0: round 1's chunking of the stride-2 layers too)
            yield
    elif path == "direct":
        with _hip.options(NO_WINOGRAD=3):
            yield
    else:
        assert _hip.get_option("NO_WINOGRAD") == 0 and _hip.get_option("NO_WINO4") == 0
        yield


def check_bf16_band(errs, band):
    """bf16 gradients of a HIP run against the fp64 evaluation of the bf16 oracle (oracle/train.py::bf16_gradient_band).
    The band (the oracle's own fp32-vs-fp64 distances) is a SAMPLE of a chaotic process over ~150 tensors and so is
    the run under test: every tensor within 2.5x the oracle's own worst, cosine >= 0.85 (or the oracle's own worst
    cosine less 0.1) - except at most 2 % of the tensors, which may reach 4x with cosine >= 0.8 (round 5: 0.6 until the
    train-mode forward got its sharp pin, tests/test_ecapa_bf16_gpu.py::test_train_mode_every_stored_tensor_teacher_forced);
    the median within 2.5x the oracle's median."""
    cos_floor = min(0.85, band["min_cos"] - 0.1)
    out = [k for k, (err, cos) in errs.items() if err > 2.5 * band["max"] or cos < cos_floor]
    print("outside the band:", [(k, round(errs[k][0], 3), round(errs[k][1], 3)) for k in out])
    assert len(out) <= max(1, len(errs) // 50), (out, band)
    for k in out:
        assert errs[k][0] <= 4.0 * band["max"] and errs[k][1] >= 0.8, (k, errs[k], band)
    assert np.median([e for e, _ in errs.values()]) <= 2.5 * band["median"], (np.median([e for e, _ in errs.values()]), band)


def record(name, value):
    """Append a measured parity figure to gpurun_out/parity_measured.jsonl (DESIGN.md quotes these)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "value": value}) + "\n")
    except OSError:
        pass


if __name__ == "__main__":
    for r in conv_budget():
        print(r)
    print("ratio (rss)", winograd_ratio(), "kernel bound", wino_conv_bound())
    for k in STRICT:
        print(k, "strict", tol(k, "strict"), "default", tol(k, "default"))
