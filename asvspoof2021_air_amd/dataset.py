"""Dataset-side API surface of the hot path (dataset.py:56-85, :513-528; main_train.py:338).

The reference's Datasets read pre-extracted ``.pt`` LFCC files of real corpora
(out of scope: SURVEY.md §2 row 7).  What the hot path needs from them is the
``feat_len`` pad/chop semantics, the crop-offset draw and the transpose; this
module keeps those, on the GPU.
"""
import numpy as np
import torch

from . import _hip
from .feature_extraction import pad_mode_id


def pad_transpose(feat, feat_len=750, start=None, padding="repeat", silence_row=None):
    """(B, T, D) GPU features -> (B, D, feat_len): pad (dataset.py:513-528: 'repeat' tiles, 'zero' appends
    zeros, 'silence' PREPENDS ``silence_row`` (D,) = LFCC.silence_row()) or chop at ``start``
    (dataset.py:68-70), then the trainer's transpose (main_train.py:338).  ``start``: optional int32 (B,)
    GPU tensor, clamped to [0, T - feat_len] on the device."""
    B, T, D = feat.shape
    mode = pad_mode_id(padding)
    if mode == 2 and silence_row is None:
        raise ValueError("padding='silence' needs the silence frame (LFCC.silence_row(device))")
    out = torch.empty((B, D, feat_len), device=feat.device, dtype=torch.float32)
    lib = _hip.lib()
    _hip.check(lib.air_pad_transpose_ex(_hip.dptr(feat), _hip.ci(B), _hip.ci(T), _hip.ci(D),
                                        _hip.dptr(out), _hip.ci(feat_len),
                                        _hip.dptr(start, torch.int32, True), _hip.ci(mode),
                                        _hip.dptr(silence_row, allow_none=True), _hip.stream()),
               "air_pad_transpose_ex")
    return out


def chop_starts(T, feat_len, batch, rng=np.random):
    """Per-utterance crop offsets with the reference's exclusive upper bound
    (dataset.py:69: ``np.random.randint(T - feat_len)``: the last valid offset is never drawn).
    None when nothing is cropped (T <= feat_len)."""
    if T <= feat_len:
        return None
    return torch.tensor([rng.randint(T - feat_len) for _ in range(batch)], dtype=torch.int32)
