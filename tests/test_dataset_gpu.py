"""Row b2 on the GPU: the Dataset surface backed by raw PCM (features from the fused HIP LFCC kernel) against the
"``.pt``-style" oracle path (oracle LFCC -> oracle pad / chop, i.e. what preprocess.py + dataset.py produce), the body
of the reference's training loop (main_train.py:310-348) and generate_score.test_on_dataset driven from a real
DataLoader, and the silence frame of ``--padding silence`` against the reference's own value."""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

import dataset_fixture as fx
from asvspoof2021_air_amd import dataset as air_ds
from oracle import lfcc as o_lfcc, pad as o_pad
from oracle.filler import fill_module_, fill_state, fill_value, synth_pcm

pytestmark = pytest.mark.gpu
LFCC_ATOL = 3e-5  # tests/test_lfcc_gpu.py's constant for the kernel against the reference goldens


def _oracle_item(pcm, feat_len, padding, silence_row):
    """(1, feat_len, 60): the reference's preprocess (LFCC of one wav) + Dataset pad / chop; draws from np.random."""
    feat = torch.from_numpy(o_lfcc.lfcc_forward(pcm.numpy()[None].copy()))
    return o_pad.pad_chop(feat, feat_len, padding, silence_row)


def _pcm_corpus(lengths, seed):
    return [synth_pcm(1, L, seed=seed + i)[0] for i, L in enumerate(lengths)]


def test_silence_padding_on_feature_files_uses_the_kernels_silence_frame(golden, tmp_path):
    g = golden("dataset.npz")
    root = fx.build(str(tmp_path))
    ds = air_ds.ASVspoof2019("LA", os.path.join(root, "la19"), "train", "LFCC", feat_len=fx.FEAT_LEN, padding="silence")
    item = ds[0][0]  # 40 frames: 56 silence frames PREPENDED (dataset.py:528)
    ref_sil = g["silence_pad_value"].reshape(-1)
    assert item.shape == (1, 96, 60)
    np.testing.assert_allclose(item[0, :56].numpy(), np.broadcast_to(ref_sil, (56, 60)), atol=LFCC_ATOL)
    np.testing.assert_array_equal(item[0, 56:].numpy(), fx.feature_of(*fx.ORI[0])[0].numpy())


@pytest.mark.parametrize("padding", ["repeat", "zero", "silence"])
def test_pcm_backed_items_equal_the_pt_style_oracle(golden, padding):
    """ASVspoof2019 over a PCMSource: ragged utterances shorter than, equal to and longer than feat_len."""
    feat_len = 60
    lengths = [3200, 9440, 9600, 12345, 16000, 24000]  # T = 21, 60, 61, 78, 101, 151
    names = ["%05d_LA_T_%07d_%s" % (i, 1000 + i, "-_bonafide" if i % 2 == 0 else "A%02d_spoof" % (1 + i)) for i in range(6)]
    wavs = _pcm_corpus(lengths, 40)
    ds = air_ds.ASVspoof2019("LA", None, "train", feat_len=feat_len, padding=padding,
                             source=air_ds.PCMSource(list(zip(names, wavs))))
    sil = torch.from_numpy(golden("dataset.npz")["silence_pad_value"]).reshape(-1)
    np.random.seed(5)
    got = [ds[i] for i in range(6)]
    np.random.seed(5)
    want = [_oracle_item(w, feat_len, padding, sil) for w in wavs]
    for i, (item, w) in enumerate(zip(got, want)):
        feat, fn, tag, lab = item
        assert feat.is_cuda and feat.shape == (1, feat_len, 60) and fn == "LA_T_%07d" % (1000 + i)
        assert (tag, lab) == ((0, 0) if i % 2 == 0 else (1 + i, 1))
        np.testing.assert_allclose(feat.cpu().numpy(), w.numpy(), atol=LFCC_ATOL, err_msg=str(i))
    # pad_chop=False: the plain (1, T, 60) LFCC
    raw = air_ds.ASVspoof2019("LA", None, "train", feat_len=feat_len, pad_chop=False, source=ds.source)[3][0]
    np.testing.assert_allclose(raw.cpu().numpy(), o_lfcc.lfcc_forward(wavs[3].numpy()[None].copy()), atol=LFCC_ATOL)


def test_batched_collate_equals_per_item_features():
    """return_pcm=True: ONE fused launch per utterance length in collate_fn == the per-item launches, bit for bit, with
    the same crop draws; the batch is the (B, 1, feat_len, 60) view whose transpose(2, 3) is contiguous."""
    feat_len = 60
    lengths = [16000, 8000, 16000, 16000, 8000, 4000]
    wavs = _pcm_corpus(lengths, 90)
    names = ["%05d_LA_D_%07d_A02_spoof" % (i, i) for i in range(6)]
    src = air_ds.PCMSource(list(zip(names, wavs)))
    per_item = air_ds.ASVspoof2019("LA", None, "dev", feat_len=feat_len, source=src)
    batched = air_ds.ASVspoof2019("LA", None, "dev", feat_len=feat_len, source=src, return_pcm=True)
    np.random.seed(3)
    a = next(iter(DataLoader(per_item, batch_size=6, shuffle=False, collate_fn=per_item.collate_fn)))
    np.random.seed(3)
    b = next(iter(DataLoader(batched, batch_size=6, shuffle=False, collate_fn=batched.collate_fn)))
    assert a[0].shape == b[0].shape == (6, 1, feat_len, 60)
    assert torch.equal(a[0], b[0]) and list(a[1]) == list(b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert b[0].transpose(2, 3).is_contiguous()


def test_training_loop_body_runs_from_a_dataloader():
    """main_train.py:310-348 verbatim on an ASVspoof2021LA_aug over PCM sources: two DataLoaders (original / augmented
    index ranges) -> unpack the 5-tuples -> cat -> ``feat.transpose(2, 3).to(device)`` -> model -> ang_iso loss; features,
    embedding and loss equal the oracle's on the .pt-style features."""
    import torch.utils.data.sampler as torch_sampler
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from oracle import resnet as o_resnet
    from oracle.loss import ocsoftmax_forward
    feat_len, n_ori, n_aug = 96, 6, 4
    ori = air_ds.SyntheticSource(688, n_ori, length=8000, part="train")
    aug = air_ds.SyntheticSource(688, n_aug, length=12000, part="train", channels=["g728", "gsmfr", "amr[br=5k9]"], first=50)
    training_set = air_ds.ASVspoof2021LA_aug(ori_source=ori, aug_source=aug, feat_len=feat_len, return_pcm=True)
    assert len(training_set) == n_ori + n_aug and len(training_set.channel) == 60
    feat, _, _, _, _ = training_set.__class__(ori_source=ori, aug_source=aug, feat_len=feat_len)[3]  # main_train.py:249
    assert tuple(feat.shape) == (1, feat_len, 60)
    trainOriDataLoader = DataLoader(training_set, batch_size=3, shuffle=False, num_workers=0, collate_fn=training_set.collate_fn,
                                    sampler=torch_sampler.SequentialSampler(range(n_ori)))
    trainAugDataLoader = DataLoader(training_set, batch_size=2, shuffle=False, num_workers=0, collate_fn=training_set.collate_fn,
                                    sampler=list(range(n_ori, n_ori + n_aug)))
    device = torch.device("cuda")
    feat_model = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(feat_model)
    feat_model.set_attention_noise(None)
    feat_model = feat_model.to(device).train()
    ang_iso = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(ang_iso)
    ang_iso = ang_iso.to(device)
    # ---- the loop body (main_train.py:310-348, :376)
    featOri, audio_fnOri, tagsOri, labelsOri, channelsOri = next(iter(trainOriDataLoader))
    featAug, audio_fnAug, tagsAug, labelsAug, channelsAug = next(iter(trainAugDataLoader))
    feat = torch.cat((featOri, featAug), 0)
    tags = torch.cat((tagsOri, tagsAug), 0)
    labels = torch.cat((labelsOri, labelsAug), 0)
    channels = torch.cat((channelsOri, channelsAug), 0)
    feat = feat.transpose(2, 3).to(device)
    tags, labels = tags.to(device), labels.to(device)
    feats, feat_outputs = feat_model(feat)
    ang_isoloss, _ = ang_iso(feats, labels)
    # ---- oracle on the .pt-style features of the same utterances
    idx = [0, 1, 2, n_ori, n_ori + 1]
    want_feat = []
    for i in idx:
        k, src, j = training_set._locate(i)
        want_feat.append(_oracle_item(src.pcm(j), feat_len, "repeat", None))
    want_feat = torch.stack(want_feat)  # (5, 1, feat_len, 60)
    assert tuple(feat.shape) == (5, 1, 60, feat_len)
    np.testing.assert_allclose(feat.cpu().numpy(), want_feat.transpose(2, 3).numpy(), atol=LFCC_ATOL)
    assert list(audio_fnOri) == ["LA_T_%07d" % (1000000 + i) for i in range(3)] and channels.tolist()[:3] == [0, 0, 0]
    assert channels.tolist()[3:] == [training_set.channel_dict["g728"], training_set.channel_dict["gsmfr"]]
    want_labels = [ori._utt(i, label_only=True)[1] for i in range(3)] + [aug._utt(50 + i, label_only=True)[1] for i in range(2)]
    assert labels.tolist() == want_labels and tags.tolist() == [0 if l == 0 else 1 + (i % 6) for i, l in zip([0, 1, 2, 50, 51], want_labels)]
    sd = fill_state(o_resnet.resnet18_shapes())
    with torch.no_grad():
        o_feats, _ = o_resnet.resnet18_forward(sd, want_feat.transpose(2, 3).contiguous(), training=True, noise=None)
    o_loss, _ = ocsoftmax_forward(o_feats, fill_value("center", (1, 256)), labels.cpu(), 0.9, 0.2, 20.0)
    # (train-mode BatchNorm on 5 utterances amplifies the 3e-5 feature differences; the golden-vector tests hold the
    # model itself to 1e-5 on identical inputs)
    np.testing.assert_allclose(feats.detach().cpu().numpy(), o_feats.numpy(), atol=2e-3 * float(o_feats.abs().max()))
    np.testing.assert_allclose(float(ang_isoloss.detach()), float(o_loss), rtol=2e-3)
    ang_isoloss.backward()  # main_train.py:406
    assert feat_model.conv1.weight.grad is not None and torch.isfinite(feat_model.conv1.weight.grad).all()


def test_generate_score_from_the_dataset(tmp_path):
    """generate_score.py:75-119 fed by DataLoader(ASVspoof2019 eval over PCM, batch_size=1) - the reference's own
    configuration (:73) - and by the batched collate: the same score file, equal to score_pcm on the waveforms."""
    from asvspoof2021_air_amd.generate_score import score_pcm, test_on_dataset
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    model = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(model)
    model.set_attention_noise(None)
    model = model.cuda()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    src = air_ds.SyntheticSource(689, 5, length=16000, part="eval")
    test_set = air_ds.ASVspoof2019("LA", None, "eval", feat_len=128, source=src)
    fa, fb = tmp_path / "a.txt", tmp_path / "b.txt"
    dl1 = DataLoader(test_set, batch_size=1, shuffle=False, num_workers=0, collate_fn=test_set.collate_fn)
    assert test_on_dataset(model, dl1, str(fa), lossm, "ocsoftmax", task="19eval", keep_dataset_labels=True) == 5
    fast = air_ds.ASVspoof2019("LA", None, "eval", feat_len=128, source=src, return_pcm=True)
    dl5 = DataLoader(fast, batch_size=5, shuffle=False, num_workers=0, collate_fn=fast.collate_fn)
    test_on_dataset(model, dl5, str(fb), lossm, "ocsoftmax", task="19eval", keep_dataset_labels=True)
    la, lb = fa.read_text().splitlines(), fb.read_text().splitlines()
    assert len(la) == len(lb) == 5
    for x, y in zip(la, lb):  # batch 1 and batch 5 pick different conv tilings: same names and keys, scores to 1e-6
        assert x.split()[0] == y.split()[0] and x.split()[2] == y.split()[2]
        assert abs(float(x.split()[1]) - float(y.split()[1])) <= 1e-6
    pcm = torch.stack([src.pcm(i) for i in range(5)]).cuda()
    want = score_pcm(model, lossm, pcm, feat_len=128).cpu().tolist()
    for i, ln in enumerate(lb):
        name, val, key = ln.split()
        assert name == "LA_E_%07d" % (1000000 + i) and key == ("spoof" if src._utt(i, label_only=True)[1] else "bonafide")
        assert float(val) == want[i]


def test_forked_dataloader_workers(golden, tmp_path):
    """the reference's Dataset is pure CPU and exposes --num_workers; this port touches the GPU.
    (a) ``.pt`` feature files with padding='silence' work under FORKED workers: the silence frame is computed once at
    construction in the parent and inherited (it used to launch the HIP LFCC inside the worker: 'Cannot re-initialize
    CUDA in forked subprocess'); items equal the in-process ones.  (b) a PCM source inside a forked worker fails with
    the remedy in the message, not with torch's."""
    from torch.utils.data import DataLoader
    root = fx.build(str(tmp_path))
    torch.zeros(1, device="cuda")  # the parent owns a GPU context before the fork
    ds = air_ds.ASVspoof2019("LA", os.path.join(root, "la19"), "train", "LFCC", feat_len=fx.FEAT_LEN, padding="silence")
    assert ds._silence_cpu is not None and not ds._silence_cpu.is_cuda
    want = [ds[i][0] for i in range(len(ds))]
    dl = DataLoader(ds, batch_size=1, shuffle=False, num_workers=2, multiprocessing_context="fork")
    got = [b[0][0] for b in dl]
    assert len(got) == len(want)
    for a in got:
        assert a.shape[-2:] == (fx.FEAT_LEN, 60)
    # (items longer than feat_len are cropped at random offsets: the padded ones are compared)
    short = [i for i in range(len(ds)) if torch.load(ds.source.path(i)).shape[1] < fx.FEAT_LEN]
    assert short
    for i in short:
        np.testing.assert_array_equal(got[i].numpy(), want[i].numpy())
    names = ["%05d_LA_T_%07d_-_bonafide" % (i, 1000 + i) for i in range(4)]
    pds = air_ds.ASVspoof2019("LA", None, "train", feat_len=60, padding="repeat",
                              source=air_ds.PCMSource(list(zip(names, _pcm_corpus([9600] * 4, 3)))))
    with pytest.raises(RuntimeError, match="num_workers=0"):
        list(DataLoader(pds, batch_size=2, num_workers=1, multiprocessing_context="fork"))


def test_synthetic_source_cache_is_bounded():
    src = air_ds.SyntheticSource(688, 40, length=4000, cache_items=8)
    for i in range(40):
        src.pcm(i)
    assert len(src._cache) == 8 and list(src._cache)[-1] == 39
    a = src.pcm(3).clone()
    assert torch.equal(a, air_ds.SyntheticSource(688, 40, length=4000, cache_items=None).pcm(3))  # regenerated identically


def test_pcm_batches_through_the_device_prefetcher_feed_the_trainer():
    """``return_pcm='batch'`` + ``collate_fn`` hand the loader's batch over as one pinned (B, L) host
    tensor; ``DevicePrefetcher`` copies batch n + 1 on a copy stream under step n.  The GPU batches are the dataset's
    waveforms and labels bit for bit, in the loader's order, over two epochs; ``Trainer.step`` on them equals the step on
    the same tensors moved by hand (fused LFCC inside the step either way); ragged lengths in one batch are refused."""
    from torch.utils.data import DataLoader
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    from oracle.filler import fill_module_
    src = air_ds.SyntheticSource(688, 12, length=8000, cache_items=None)
    ds = air_ds.ASVspoof2019("LA", None, "train", feat_len=96, padding="repeat", source=src, return_pcm="batch")
    dl = DataLoader(ds, batch_size=4, shuffle=False, collate_fn=ds.collate_fn, num_workers=0)
    want = [(torch.stack([src.pcm(4 * b + j) for j in range(4)]), [ds[4 * b + j][3] for j in range(4)]) for b in range(3)]
    host = next(iter(dl))
    assert host[0].is_pinned() and host[0].shape == (4, 8000) and not host[0].is_cuda
    for epoch in range(2):
        got = list(air_ds.DevicePrefetcher(dl, "cuda", depth=2))
        assert len(got) == 3
        for (pcm, names, tags, labels), (wp, wl) in zip(got, want):
            assert pcm.is_cuda and labels.is_cuda and torch.equal(pcm.cpu(), wp) and labels.cpu().tolist() == wl
            assert isinstance(names, (list, tuple)) and names[0].startswith("LA_T_")

    def make():
        m = ResNet(3, 256, resnet_type="18", nclasses=2)
        fill_module_(m)
        m.set_attention_noise(None)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        return Trainer(m, loss_module=lossm, feat_len=96)

    a, b = make(), make()
    for (pcm, _, _, labels), (wp, wl) in zip(air_ds.DevicePrefetcher(dl, "cuda"), want):
        la = a.step(pcm, labels)[0]
        lb = b.step(wp.cuda(), torch.tensor(wl).cuda())[0]
        assert torch.equal(la, lb)
    assert torch.equal(a.model.arena().flat, b.model.arena().flat)
    ragged = air_ds.ASVspoof2019("LA", None, "train", feat_len=96, return_pcm="batch", source=air_ds.PCMSource(
        [("%05d_LA_T_%07d_-_bonafide" % (i, i), w) for i, w in enumerate(_pcm_corpus([8000, 8160], 3))]))
    with pytest.raises(ValueError, match="one length"):
        ragged.collate_fn([ragged[0], ragged[1]])
