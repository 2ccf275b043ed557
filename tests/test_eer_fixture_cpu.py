"""CPU: the committed run-to-run distribution of the 4 s ResNet recipe's final-epoch loss (tests/golden/
eer_chaos_4s_resnet.json: 20 runs on the all-f32 kernels, 20 on the split-bf16 stride-2 kernels, round 5) and the gate
tests/test_eer_gpu.py derives from it (the evidence was prose in profiles/).

What the fixture must show for the gate to mean anything:
  * the two arithmetics are samples of ONE distribution (medians within 2 %, a rank test does not separate them), and
    that distribution sits on the reference's own final loss;
  * the gate's false-alarm rate on that distribution is small and STATED: 'at most one of three above the 92.5th
    percentile' fails a correct build with probability 3 p^2 (1 - p) + p^3 at p = P(sample > tail);
  * a regression of the kind ADVICE names (one arithmetic path ending runs at 0.15 - 0.30) fails it."""
import numpy as np
from scipy import stats


def _gate():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("test_eer_gpu", os.path.join(os.path.dirname(__file__), "test_eer_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.chaos_gate()


def _passes(g, floors):
    return (sum(f > g["tail"] for f in floors) <= 1 and max(floors) < g["cap"] and float(np.median(floors)) <= g["median_hi"])


def test_two_arithmetics_one_distribution():
    g = _gate()
    a, b = g["arms"]["f32 (CONV_S2=3)"], g["arms"]["split-bf16 (CONV_S2=31)"]
    assert a.size == 20 and b.size == 20
    assert abs(np.median(a) - np.median(b)) <= 0.02 * np.median(a)
    assert stats.mannwhitneyu(a, b, alternative="two-sided").pvalue > 0.2
    assert abs(np.median(g["values"]) - g["reference"]) <= 0.05 * g["reference"]
    # the tail: 3 of 40 runs still on a transient at the last epoch
    assert int((g["values"] > g["tail"]).sum()) == 3 and 0.10 < g["tail"] < 0.12 and 0.30 < g["cap"] < 0.32


def test_gate_false_alarm_rate_is_stated():
    g = _gate()
    p = float((g["values"] > g["tail"]).mean())
    q = float((g["values"] > g["median_hi"]).mean())
    fa_tail = 3 * p * p * (1 - p) + p ** 3
    fa_med = 3 * q * q * (1 - q) + q ** 3  # (the median of three exceeds x iff two of them do)
    assert fa_tail <= 0.02 and fa_med <= 0.03, (fa_tail, fa_med)
    # exhaustively over the 40^3 ordered triples of the empirical distribution: what the GPU test would do on a correct build
    v = g["values"]
    fails = sum(0 if _passes(g, (x, y, z)) else 1 for x in v for y in v for z in v)
    assert fails / v.size ** 3 <= 0.03, fails / v.size ** 3


def test_gate_catches_a_shifted_floor():
    g = _gate()
    rng = np.random.default_rng(0)
    # a build whose runs end at 0.15 - 0.30 ("a real regression of one arithmetic path up to 0.30 final loss
    # would now pass"): every triple drawn from that range must fail
    bad = rng.uniform(0.15, 0.30, size=(200, 3))
    assert not any(_passes(g, tuple(t)) for t in bad)
    # ... and a milder one - half the runs at the floor, half at 0.12 - 0.2 - fails most of the time
    mild = np.where(rng.random((2000, 3)) < 0.5, rng.choice(g["values"][g["values"] < 0.09], size=(2000, 3)),
                    rng.uniform(0.12, 0.2, size=(2000, 3)))
    assert np.mean([not _passes(g, tuple(t)) for t in mild]) >= 0.45
