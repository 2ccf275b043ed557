"""Separable synthetic anti-spoofing corpus for the EER parity check.

White noise carries no class information, so the held-out EER check uses a tiny
speech-like corpus: bona fide = harmonic stack (f0 ~ U[90,250] Hz, 1/k roll-off) shaped by a
random two-pole formant resonance plus noise at 25 dB SNR; spoof = the same generator with a
vocoder-like artefact (a spectral notch near 3.2 kHz and a faint periodic buzz at 4 kHz, both of random strength down to zero, so the classes overlap).
Deterministic per (seed, index); pure numpy so the reference/oracle (CPU) and the HIP path
see bit-identical PCM.
"""
import numpy as np


def utterance(seed, idx, length=32000, sr=16000, mix_lo=0.0):
    """Returns (pcm float32 (length,), label) with label 0 = bona fide, 1 = spoof."""
    rng = np.random.Generator(np.random.PCG64(seed * 1000003 + idx))
    label = int(rng.random() < 0.5)
    t = np.arange(length) / sr
    f0 = rng.uniform(90.0, 250.0) * (1.0 + 0.02 * np.sin(2 * np.pi * rng.uniform(2, 6) * t))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(length)
    for k in range(1, 31):
        x += np.sin(k * phase + rng.uniform(0, 2 * np.pi)) / k
    # two-pole formant resonance
    fc, bw = rng.uniform(500.0, 2500.0), rng.uniform(80.0, 300.0)
    r = np.exp(-np.pi * bw / sr)
    a1, a2 = -2 * r * np.cos(2 * np.pi * fc / sr), r * r
    # y[n] = x[n] - a1 y[n-1] - a2 y[n-2] (round 3: scipy's filter instead of a Python loop over the samples -
    # the 4096-utterance held-out sets and the 4 s variant need ~20 M samples)
    from scipy.signal import lfilter
    y = lfilter([1.0], [1.0, a1, a2], x)
    y /= np.abs(y).max() + 1e-9
    if label == 1:
        # notch around 3.2 kHz (second-order zero pair) + 4 kHz buzz
        fz = 3200.0 + rng.uniform(-100, 100)
        b1 = -2 * np.cos(2 * np.pi * fz / sr)
        z = y.copy()
        z[2:] = y[2:] + b1 * y[1:-1] + y[:-2]
        z = z / (np.abs(z).max() + 1e-9)
        # cue strength varies per utterance, down to none: the classes overlap, so the
        # converged EER is a stable non-zero number instead of 0
        mix = rng.uniform(mix_lo, 0.8)
        y = (1.0 - mix) * y + mix * z
        y = y + 0.01 * mix * np.sign(np.sin(2 * np.pi * 4000.0 * t))
    noise = rng.standard_normal(length)
    y = y + noise * (np.sqrt(np.mean(y ** 2)) / np.sqrt(np.mean(noise ** 2)) * 10 ** (-25 / 20))
    y = 0.3 * y / (np.abs(y).max() + 1e-9)
    return y.astype(np.float32), label


def corpus(seed, n, length=32000, mix_lo=0.0):
    """(pcm (n, length) float32, labels (n,) int64)."""
    pcm = np.zeros((n, length), dtype=np.float32)
    labels = np.zeros(n, dtype=np.int64)
    for i in range(n):
        pcm[i], labels[i] = utterance(seed, i, length, mix_lo=mix_lo)
    return pcm, labels
