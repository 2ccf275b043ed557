"""Oracle (test infrastructure): feature pad / chop to ``feat_len`` frames.

Restates the block every reference Dataset shares (dataset.py:66-79) and the
three pad helpers (dataset.py:513-528), plus the trainer's transpose
(main_train.py:338, :347-348).
"""
import numpy as np
import torch


def zero_pad(spec, ref_len):
    """dataset.py:513-517: append zero frames."""
    _, cur, width = spec.shape
    assert ref_len > cur
    return torch.cat((spec, torch.zeros((1, ref_len - cur, width), dtype=spec.dtype)), 1)


def repeat_pad(spec, ref_len):
    """dataset.py:519-522: tile along time then cut: frame t -> frame t mod T."""
    mul = int(np.ceil(ref_len / spec.shape[1]))
    return spec.repeat(1, mul, 1)[:, :ref_len, :]


def silence_pad(spec, ref_len, silence_row):
    """dataset.py:524-528: PREPENDS the silence frame (LFCC of zeros, dataset.py:13-16)."""
    _, cur, width = spec.shape
    assert ref_len > cur
    return torch.cat((silence_row.reshape(1, 1, width).repeat(1, ref_len - cur, 1), spec), 1)


def pad_chop(spec, feat_len=750, padding="repeat", silence_row=None, rng=np.random):
    """dataset.py:66-79.  spec: (1, T, D).  Longer inputs are cropped at
    ``rng.randint(T - feat_len)`` (exclusive upper bound: last offset never drawn)."""
    T = spec.shape[1]
    if T > feat_len:
        start = rng.randint(T - feat_len)
        return spec[:, start:start + feat_len, :]
    if T < feat_len:
        if padding == "zero":
            return zero_pad(spec, feat_len)
        if padding == "repeat":
            return repeat_pad(spec, feat_len)
        if padding == "silence":
            return silence_pad(spec, feat_len, silence_row)
        raise ValueError("Padding should be zero or repeat!")
    return spec


def to_model_input(feat, ecapa=False):
    """main_train.py:338 (+ :347-348 for ECAPA).  feat: (B, 1, T, D) ->
    (B, 1, D, T) view, squeezed to (B, D, T) for ECAPA."""
    feat = feat.transpose(2, 3)
    return torch.squeeze(feat) if ecapa else feat
