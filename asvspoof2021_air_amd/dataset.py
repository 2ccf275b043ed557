"""Dataset-side API surface of the hot path (dataset.py:56-85, :513-528; main_train.py:338).

The reference's Datasets read pre-extracted ``.pt`` LFCC files of real corpora
(out of scope: SURVEY.md §2 row 7).  What the hot path needs from them is the
tuple layout, ``feat_len`` pad/chop semantics and the transpose; this module
keeps those, on the GPU.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import _hip


def pad_transpose(feat, feat_len=750, start=None):
    """(B, T, D) GPU features -> (B, D, feat_len): repeat-pad (dataset.py:519-522)
    or chop at ``start`` (dataset.py:68-70), then the trainer's transpose
    (main_train.py:338).  ``start``: optional int32 (B,) GPU tensor."""
    B, T, D = feat.shape
    out = torch.empty((B, D, feat_len), device=feat.device, dtype=torch.float32)
    lib = _hip.lib()
    _hip.check(lib.air_pad_transpose(_hip.dptr(feat), _hip.ci(B), _hip.ci(T), _hip.ci(D),
                                     _hip.dptr(out), _hip.ci(feat_len),
                                     _hip.dptr(start, torch.int32, True), _hip.stream()),
               "air_pad_transpose")
    return out


def chop_starts(T, feat_len, batch, rng=np.random):
    """Per-utterance crop offsets with the reference's exclusive upper bound
    (dataset.py:69: ``np.random.randint(T - feat_len)``)."""
    if T <= feat_len:
        return None
    return torch.tensor([rng.randint(T - feat_len) for _ in range(batch)], dtype=torch.int32)


class SyntheticPCM(Dataset):
    """Synthetic raw-audio Dataset with the reference's item layout
    ``(waveform:(1,L) f32 in [-1,1] @16 kHz, filename, tag, label)`` (raw_dataset.py:27,66).

    Labels follow ASVspoof2019 LA's ~90 % spoof prior (dataset.py:43)."""

    def __init__(self, n=2048, length=64000, seed=688, p_spoof=0.9):
        self.n, self.length, self.seed = n, length, seed
        g = torch.Generator().manual_seed(seed)
        self.labels = (torch.rand(n, generator=g) < p_spoof).long()
        self.labels[0] = 0
        self.labels[-1] = 1

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 100003 + idx)
        wav = 0.1 * torch.randn(1, self.length, generator=g)
        label = int(self.labels[idx])
        return wav, "SYN_%07d" % idx, (1 if label else 0), label
