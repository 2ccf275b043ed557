"""Row b2 of the coverage table: the Dataset surface.  This package's Dataset classes on a small corpus of
``.pt`` feature files against the REAL reference classes run on the same files (tests/golden/dataset.npz, written by
tests/golden/make_golden_dataset.py): every item of every class - features, filename, tag, label, channel / device -
over two passes of pinned crop draws, and the DataLoader batches through each class's collate_fn."""
import ast

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

import dataset_fixture as fx
from asvspoof2021_air_amd import dataset as air_ds


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    return fx.build(str(tmp_path_factory.mktemp("corpus")))


def _same_digest(g, prefix, feat):
    d = fx.digest(feat.numpy())
    assert tuple(d["shape"]) == tuple(g[prefix + "/shape"]), prefix
    for key in ("sums", "first", "last"):
        np.testing.assert_array_equal(d[key], g[prefix + "/" + key], err_msg=prefix + "/" + key)


def _plain(v):
    return v if isinstance(v, str) else np.asarray(v).tolist()


@pytest.mark.parametrize("name", ["la19_repeat", "la19_zero", "la19_silence", "la19_eval", "pa19", "la19_nopad", "aug_la",
                                  "aug_df", "aug_lapa", "aug_dfpa", "eval21_la", "eval21_df"])
def test_items_and_batches_equal_the_reference(golden, corpus, name):
    g = golden("dataset.npz")
    ds = fx.cases(air_ds, corpus)[name]()
    if "silence" in name or name == "eval21_df":
        # the silence frame is an LFCC of zeros (dataset.py:13-16): the GPU test computes it with the HIP kernel, here it
        # is the reference's own value
        sil = torch.from_numpy(g["silence_pad_value"]).reshape(-1)
        ds._silence_row = lambda like: sil
    order = [int(i) for i in g[name + "/order"]]
    assert len(ds) * 2 == len(order)
    want = ast.literal_eval(str(g[name + "/meta"]))
    np.random.seed(1234)
    for k, i in enumerate(order):
        item = ds[i]
        assert isinstance(item, tuple) and len(item) == 1 + len(want[k])
        _same_digest(g, "%s/feat%d" % (name, k), item[0])
        assert [_plain(v) for v in item[1:]] == want[k], (name, k)
        # the types main_train.py:312's default_collate relies on
        assert isinstance(item[1], str)
        if len(item) > 2:
            assert isinstance(item[2], int) and isinstance(item[3], int)
    np.random.seed(99)
    dl = DataLoader(ds, batch_size=3, shuffle=False, num_workers=0, collate_fn=ds.collate_fn)
    nb = 0
    for bi, batch in enumerate(dl):
        _same_digest(g, "%s/batch%d_feat" % (name, bi), batch[0])
        rest = [(list(b) if isinstance(b, (list, tuple)) else b.tolist()) for b in batch[1:]]
        assert rest == ast.literal_eval(str(g["%s/batch%d_rest" % (name, bi)])), (name, bi)
        nb += 1
    assert nb == -(-len(ds) // 3)


def test_maps_and_errors():
    """tag / label / channel / device maps (dataset.py:31-38, :121-141, :215-219) and the error behaviour."""
    assert air_ds.TAG_LA19["A19"] == 19 and air_ds.TAG_PA19 == {"-": 0, "AA": 1, "AB": 2, "AC": 3, "BA": 4, "BB": 5, "BC": 6,
                                                               "CA": 7, "CB": 8, "CC": 9}
    assert len(air_ds.CHANNELS_LA) == 60 and air_ds.CHANNELS_LA[0] == "no_channel" and air_ds.CHANNELS_LA[38] == "gsmfr"
    assert air_ds.CHANNELS_DF == ["no_channel", "aac[16k]", "aac[32k]", "aac[8k]", "mp3[16k]", "mp3[32k]", "mp3[8k]"]
    assert len(air_ds.DEVICES) == 13 and air_ds.DEVICES[-1] == ""
    with pytest.raises(ValueError, match="Access type should be LA or PA!"):
        air_ds.ASVspoof2019("XX", "/nowhere")


def test_bad_padding_and_field_count(corpus, tmp_path):
    import os
    ds = air_ds.ASVspoof2019("LA", os.path.join(corpus, "la19"), "train", "LFCC", feat_len=96, padding="reflect")
    with pytest.raises(ValueError, match="Padding should be zero or repeat!"):
        ds[0]  # (40 frames < feat_len: the pad branch, dataset.py:79)
    ds[1]      # 96 frames: nothing to pad, no error - like the reference
    fx.write(str(tmp_path / "x" / "train" / "LFCC"), [("00000_LA_T_1000001_bonafide", 30)])
    with pytest.raises(AssertionError):
        air_ds.ASVspoof2019("LA", str(tmp_path / "x"), "train", "LFCC")[0]


def test_genuine_only(corpus):
    import os
    la = os.path.join(corpus, "la19")
    old = dict(air_ds.ASVspoof2019.NUM_BONAFIDE)
    try:
        air_ds.ASVspoof2019.NUM_BONAFIDE = {"train": 2, "dev": 2, "eval": 2}
        tr = air_ds.ASVspoof2019("LA", la, "train", "LFCC", feat_len=96, genuine_only=True)
        assert len(tr) == 2 and all(tr[i][3] == 0 for i in range(2))   # the first N files (dataset.py:42-44)
        ev = air_ds.ASVspoof2019("LA", la, "eval", "LFCC", feat_len=96, genuine_only=True)
        assert len(ev) == 2 and all("bonafide" in f for f in ev.all_files)
        air_ds.ASVspoof2019.NUM_BONAFIDE["eval"] = 7355
        with pytest.raises(AssertionError):  # dataset.py:51 asserts the corpus size
            air_ds.ASVspoof2019("LA", la, "eval", "LFCC", feat_len=96, genuine_only=True)
    finally:
        air_ds.ASVspoof2019.NUM_BONAFIDE = old


def test_synthetic_source_names_parse():
    """SyntheticSource names follow the corpus scheme (no GPU needed to list / parse them)."""
    src = air_ds.SyntheticSource(688, 12, length=8000, part="dev", device="cpu")
    ds = air_ds.ASVspoof2019("LA", None, "dev", source=src, feat_len=50)
    assert len(ds) == 12
    for i in range(12):
        info = ds._fields(0, src.path(i))
        fn, tag, lab = ds._meta(0, info)
        assert fn.startswith("LA_D_") and lab == src._utt(src._idx[i], label_only=True)[1] and (tag == 0) == (lab == 0)
    aug = air_ds.SyntheticSource(688, 6, length=8000, channels=air_ds.CHANNELS_LA[1:4], device="cpu", first=100)
    d2 = air_ds.ASVspoof2021LA_aug(ori_source=src, aug_source=aug, feat_len=50)
    assert len(d2) == 18 and d2._meta(1, d2._fields(1, aug.path(4)))[3] == 2
