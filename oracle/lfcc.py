"""Oracle (test infrastructure): LFCC front-end restated in numpy.

Follows the reference's ``LFCC`` module step by step:

* filterbank ........ feature_extraction.py:77-86 (``trimf`` :16-39)
* DCT-II matrix ..... utils_dsp.py:147-176 (``dct``), :233-244 (``LinearDCT``)
* forward ........... feature_extraction.py:93-138
* delta ............. feature_extraction.py:41-58

``dtype`` selects the arithmetic width: float32 mirrors the reference,
float64 is the independent high-precision check used to bound the error of
both the reference and the HIP kernel.
"""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)  # torch.finfo(torch.float32).eps, feature_extraction.py:117


def linear_filterbank(fn=512, sr=16000, filter_num=20, dtype=np.float32):
    """(fn//2+1, filter_num) triangular filterbank.

    feature_extraction.py:77-86 builds ``f = sr/2 * linspace(0,1,fn//2+1)`` and
    ``filter_num+2`` linearly spaced band edges, then calls ``trimf`` (:16-39)
    per filter: rising slope on a<x<b, falling slope on b<x<c (strict
    inequalities) and exactly 1 where x == b.
    """
    nbin = fn // 2 + 1
    f = (dtype(sr) / dtype(2)) * np.linspace(0, 1, nbin).astype(dtype)
    bands = np.linspace(f.min(), f.max(), filter_num + 2).astype(dtype)
    fb = np.zeros((nbin, filter_num), dtype=dtype)
    for j in range(filter_num):
        a, b, c = bands[j], bands[j + 1], bands[j + 2]
        y = np.zeros(nbin, dtype=dtype)
        if a < b:
            m = (a < f) & (f < b)
            y[m] = (f[m] - a) / (b - a)
        if b < c:
            m = (b < f) & (f < c)
            y[m] = (c - f[m]) / (c - b)
        y[f == b] = 1
        fb[:, j] = y
    return fb


def dct2_ortho_matrix(n=20, dtype=np.float32):
    """Weight of ``LinearDCT(n, 'dct', norm='ortho')`` (utils_dsp.py:233-244).

    The reference obtains it as ``dct(eye(n), 'ortho').t()`` (utils_dsp.py:147-176);
    the closed form is D[k, m] = s_k cos(pi (2m+1) k / 2n), s_0 = sqrt(1/n),
    s_k = sqrt(2/n); ``y = x @ D.T`` (nn.Linear, no bias).
    """
    k = np.arange(n, dtype=np.float64)[:, None]
    m = np.arange(n, dtype=np.float64)[None, :]
    d = np.cos(np.pi * (2 * m + 1) * k / (2 * n))
    d[0] *= np.sqrt(1.0 / n)
    d[1:] *= np.sqrt(2.0 / n)
    return d.astype(dtype)


def hamming_periodic(fl=320, dtype=np.float32):
    """torch.hamming_window(fl) default (periodic=True): feature_extraction.py:110."""
    n = np.arange(fl, dtype=np.float64)
    return (0.54 - 0.46 * np.cos(2.0 * np.pi * n / fl)).astype(dtype)


def delta(x):
    """feature_extraction.py:41-58: out[t] = x[min(t+1,T-1)] - x[max(t-1,0)] over axis 1."""
    xp = np.concatenate([x[:, :1], x, x[:, -1:]], axis=1)
    return xp[:, 2:] - xp[:, :-2]


def pre_emphasis_(x, coef=0.97):
    """In-place, non-recursive FIR (feature_extraction.py:105-106): RHS is
    materialised from the ORIGINAL samples before the assignment."""
    x[:, 1:] = x[:, 1:] - x.dtype.type(coef) * x[:, :-1]
    return x


def lfcc_forward(x, fl=320, fs=160, fn=512, sr=16000, filter_num=20,
                 with_energy=False, with_emphasis=True, with_delta=True,
                 fb=None, dct=None, dtype=np.float32, mutate=True):
    """LFCC.forward (feature_extraction.py:93-138).

    x: (B, L) float array.  With ``mutate`` (default) the caller's array is
    pre-emphasised in place exactly like the reference (:106).
    Returns (B, 1 + L//fs, 3*filter_num) (or filter_num without deltas).
    """
    x = np.asarray(x)
    work = x if (mutate and x.dtype == dtype) else x.astype(dtype, copy=True)
    if with_emphasis:
        pre_emphasis_(work)
        if mutate and work is not x:
            x[...] = work.astype(x.dtype)
    B, L = work.shape
    T = 1 + L // fs
    # torch.stft(center=True, pad_mode="constant"): fn//2 zeros each side (:109-111)
    padded = np.zeros((B, L + fn), dtype=dtype)
    padded[:, fn // 2: fn // 2 + L] = work
    win = np.zeros(fn, dtype=dtype)
    off = (fn - fl) // 2
    win[off:off + fl] = hamming_periodic(fl, dtype)
    idx = (np.arange(T) * fs)[:, None] + np.arange(fn)[None, :]
    frames = padded[:, idx] * win  # (B, T, fn)
    spec = np.fft.rfft(frames, axis=-1)
    # :113 computes norm(.,2,-1).pow(2): sqrt then square
    amp = np.sqrt(spec.real.astype(dtype) ** 2 + spec.imag.astype(dtype) ** 2)
    sp_amp = (amp * amp).astype(dtype)  # (B, T, nbin)
    if fb is None:
        fb = linear_filterbank(fn, sr, filter_num, dtype)
    if dct is None:
        dct = dct2_ortho_matrix(filter_num, dtype)
    fb_feature = np.log10(sp_amp @ fb.astype(dtype) + dtype(EPS32)).astype(dtype)  # :116-117
    lfcc = (fb_feature @ dct.astype(dtype).T).astype(dtype)  # :120
    if with_energy:  # :123-127 (dead on the hot path; kept for completeness)
        energy = np.log10((sp_amp / dtype(fn)).sum(axis=2) + dtype(EPS32))
        lfcc[:, :, 0] = energy
    if with_delta:  # :130-133
        d1 = delta(lfcc)
        d2 = delta(d1)
        return np.concatenate([lfcc, d1, d2], axis=2).astype(dtype)
    return lfcc
