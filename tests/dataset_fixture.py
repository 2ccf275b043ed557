"""Shared by tests/golden/make_golden_dataset.py (which runs the REAL reference Dataset classes on these files) and
tests/test_dataset*.py (which run this package's classes on the same files): a small on-disk corpus of ``.pt``
feature files in the reference's naming scheme (preprocess.py:85, :156, :243), contents from a seeded formula."""
import os

import torch

from oracle.filler import synth_feat

FEAT_LEN = 96
# (name without extension, frames): shorter than, equal to and longer than FEAT_LEN
ORI = [("00000_LA_T_1000001_-_bonafide", 40), ("00001_LA_T_1000002_-_bonafide", 96), ("00002_LA_T_1000003_A01_spoof", 130),
       ("00003_LA_T_1000004_A06_spoof", 33), ("00004_LA_T_1000005_A03_spoof", 171)]
PA = [("00000_PA_T_0000001_-_bonafide", 50), ("00001_PA_T_0000002_AA_spoof", 120), ("00002_PA_T_0000003_CB_spoof", 96),
      ("00003_PA_T_0000004_BC_spoof", 70)]
EVAL19 = [("00000_LA_E_2000001_-_bonafide", 61), ("00001_LA_E_2000002_A19_spoof", 140), ("00002_LA_E_2000003_A07_spoof", 96),
          ("00003_LA_E_2000004_-_bonafide", 20)]
AUG_LA = [("000000_LA_T_1000001_-_bonafide_amr[br=5k9]", 45), ("000001_LA_T_1000003_A01_spoof_g711[law=u]", 133),
          ("000002_LA_T_1000004_A06_spoof_silkwb[br=30k,loss=5]", 96), ("000003_LA_T_1000005_A03_spoof_gsmfr", 77)]
AUG_DF = [("000000_LA_T_1000001_-_bonafide_mp3[16k]", 45), ("000001_LA_T_1000003_A01_spoof_aac[8k]", 133),
          ("000002_LA_T_1000004_A06_spoof_mp3[32k]", 96)]
AUG_LAPA = [("000000_LA_T_1000001_-_bonafide_amr[br=5k9]_iPhoneirRecording-16000.ir", 45),
            ("000001_LA_T_1000003_A01_spoof_g728_Doremi-16000.ir", 133),
            ("000002_LA_T_1000004_A06_spoof_silk[br=15k]_telephone90sC-16000.ir", 96)]
AUG_DFPA = [("000000_LA_T_1000001_-_bonafide_aac[16k]_ResloSR1-16000.ir", 45),
            ("000001_LA_T_1000003_A01_spoof_mp3[8k]_OktavaML19-16000.ir", 133)]
EVAL21 = [("000000_LA_E_9000001", 52), ("000001_LA_E_9000002", 96), ("000002_LA_E_9000003", 151)]


def feature_of(name, frames):
    seed = sum(ord(c) * (i + 1) for i, c in enumerate(name)) % 100003
    return synth_feat((1, frames, 60), seed=seed)


def write(folder, items):
    os.makedirs(folder, exist_ok=True)
    for name, frames in items:
        torch.save(feature_of(name, frames), os.path.join(folder, name + ".pt"))


def build(root):
    """root/la19/{train,eval}/LFCC, root/pa19/train/LFCC, root/aug_*/train/LFCC, root/eval21/LFCC."""
    write(os.path.join(root, "la19", "train", "LFCC"), ORI)
    write(os.path.join(root, "la19", "eval", "LFCC"), EVAL19)
    write(os.path.join(root, "pa19", "train", "LFCC"), PA)
    write(os.path.join(root, "aug_la", "train", "LFCC"), AUG_LA)
    write(os.path.join(root, "aug_df", "train", "LFCC"), AUG_DF)
    write(os.path.join(root, "aug_lapa", "train", "LFCC"), AUG_LAPA)
    write(os.path.join(root, "aug_dfpa", "train", "LFCC"), AUG_DFPA)
    write(os.path.join(root, "eval21", "LFCC"), EVAL21)
    return root


def cases(mod, root):
    """name -> constructor thunk, for a module that exposes the reference's class names."""
    la, j = os.path.join(root, "la19"), os.path.join
    out = {}
    for pad in ("repeat", "zero", "silence"):
        out["la19_" + pad] = lambda pad=pad: mod.ASVspoof2019("LA", la, "train", "LFCC", feat_len=FEAT_LEN, padding=pad)
    out["la19_eval"] = lambda: mod.ASVspoof2019("LA", la, "eval", "LFCC", feat_len=FEAT_LEN)
    out["pa19"] = lambda: mod.ASVspoof2019("PA", j(root, "pa19"), "train", "LFCC", feat_len=FEAT_LEN)
    out["la19_nopad"] = lambda: mod.ASVspoof2019("LA", la, "train", "LFCC", feat_len=FEAT_LEN, pad_chop=False)
    out["aug_la"] = lambda: mod.ASVspoof2021LA_aug(la, j(root, "aug_la"), "train", "LFCC", feat_len=FEAT_LEN)
    out["aug_df"] = lambda: mod.ASVspoof2021DF_aug(la, j(root, "aug_df"), "train", "LFCC", feat_len=FEAT_LEN, padding="zero")
    out["aug_lapa"] = lambda: mod.ASVspoof2021LAPA_aug(la, j(root, "aug_lapa"), "train", "LFCC", feat_len=FEAT_LEN)
    out["aug_dfpa"] = lambda: mod.ASVspoof2021DFPA_aug(la, j(root, "aug_dfpa"), "train", "LFCC", feat_len=FEAT_LEN)
    out["eval21_la"] = lambda: mod.ASVspoof2021LAeval(j(root, "eval21"), "LFCC", feat_len=FEAT_LEN)
    out["eval21_df"] = lambda: mod.ASVspoof2021DFeval(j(root, "eval21"), "LFCC", feat_len=FEAT_LEN, padding="silence")
    return out


def digest(feat):
    """Compact, order-sensitive fingerprint of a feature tensor (.., T, 60) (the features are seeded noise, which does
    not compress; the outputs are rearrangements of known inputs): shape, three float64 position-weighted sums, and
    the first and last frame of every leading index."""
    import numpy as np
    a = np.asarray(feat, dtype=np.float64)
    T, D = a.shape[-2], a.shape[-1]
    lead = a.reshape(-1, T, D)
    wt = np.cos(0.7310585 * np.arange(T) + 0.3)[:, None]
    wd = np.sin(1.6180339 * np.arange(D) + 0.1)[None, :]
    sums = np.stack([lead.sum((1, 2)), (lead * wt).sum((1, 2)), (lead * wt * wd).sum((1, 2))], 1)
    return {"shape": np.array(a.shape), "sums": sums, "first": np.asarray(feat).reshape(-1, T, D)[:, 0].copy(),
            "last": np.asarray(feat).reshape(-1, T, D)[:, -1].copy()}
