"""Oracle (test infrastructure): on-the-fly channel augmentation.

PARITY UNPINNED.  The reference augments OFFLINE by shelling out to
``./degrade-audio-safe-random.py -c irdevice[filter=NAME]`` from idiap/acoustic-simulator
(channel_simulation/simulated_device.py:33-35,46-50,57-61; simulated_device_channel.py:53-56).
Neither that tool nor its ``.ir`` files are under /root/reference and no test of the reference
pins their arithmetic, so this file states the spec the build implements:

    y = (x * h)[:L]                    linear convolution, truncated to the input length
    y <- y * max|x| / max|y|           "safe": peak level preserved, never clips (normalize=True)

checked here against ``scipy.signal.fftconvolve`` in float64.  IR choice per utterance follows the
reference's ``random.choice(recDevices)`` (simulated_device.py:31) with a seeded generator.
"""
import numpy as np


def ir_convolve(x, irs, idx, normalize=True):
    """x (B, L) float; irs (n_ir, H); idx (B,) ints, < 0 = unchanged.  Returns float64 (B, L)."""
    from scipy.signal import fftconvolve
    x = np.asarray(x, dtype=np.float64)
    out = x.copy()
    for b in range(x.shape[0]):
        if idx[b] < 0:
            continue
        y = fftconvolve(x[b], np.asarray(irs[idx[b]], dtype=np.float64))[: x.shape[1]]
        if normalize:
            py = np.abs(y).max()
            if py > 0:
                y = y * (np.abs(x[b]).max() / py)
        out[b] = y
    return out


def ir_convolve_direct(x, h):
    """Definition-level check of the convolution itself (small sizes): y[n] = sum_k h[k] x[n-k]."""
    x = np.asarray(x, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    y = np.zeros_like(x)
    for n in range(x.size):
        k = np.arange(0, min(n, h.size - 1) + 1)
        y[n] = np.dot(h[k], x[n - k])
    return y
