"""Scoring path (generate_score.py:75-119): score-file text on CPU, batched eval scoring on GPU."""
import numpy as np
import pytest
import torch

from oracle.filler import fill_module_, fill_state, fill_value, synth_feat


def test_score_line_format_matches_reference_text():
    """generate_score.py:113-119 formats a Python float with %s (repr of the fp32 value widened
    to double).  Known answers written out by hand from that format string."""
    from asvspoof2021_air_amd.generate_score import format_score_line
    v = torch.tensor([0.123456789, -1.0, 1e-8], dtype=torch.float32)
    vals = v.tolist()
    assert vals[0] == v[0].item()  # tolist() widens exactly like .item()
    assert format_score_line("LA_E_1000147", vals[0], "bonafide") == "LA_E_1000147 0.12345679104328156 bonafide\n"
    assert format_score_line("LA_E_1000273", vals[1], "spoof") == "LA_E_1000273 -1.0 spoof\n"
    assert format_score_line("DF_E_2000011", vals[2]) == "DF_E_2000011 9.99999993922529e-09\n"
    assert format_score_line("x", vals[0], "bonafide") == "%s %s %s\n" % ("x", v[0].item(), "bonafide")


def _items(n, feat_len, with_labels):
    feats = synth_feat((n, 1, feat_len, 60), seed=11)
    names = ["LA_E_%07d" % (1000 + i) for i in range(n)]
    labels = torch.tensor([i % 3 == 0 for i in range(n)]).long()
    return feats, names, labels


@pytest.mark.gpu
@pytest.mark.parametrize("add_loss", [None, "ocsoftmax"])
def test_batched_scoring_matches_oracle_and_batch1(tmp_path, add_loss):
    from asvspoof2021_air_amd.generate_score import test_on_dataset
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from oracle import resnet as o_resnet
    from oracle.loss import ocsoftmax_forward
    model = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(model)
    model.set_attention_noise(None)
    model = model.cuda()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    n, T = 6, 96
    feats, names, labels = _items(n, T, True)
    tags = torch.zeros(n)
    batched = [(feats[i:i + 3], names[i:i + 3], tags[i:i + 3], labels[i:i + 3]) for i in range(0, n, 3)]
    single = [(feats[i:i + 1], names[i:i + 1], tags[i:i + 1], labels[i:i + 1]) for i in range(n)]
    fa, fb, fc = tmp_path / "a" / "score.txt", tmp_path / "b.txt", tmp_path / "c.txt"
    assert test_on_dataset(model, batched, str(fa), lossm, add_loss, task="19eval") == n
    test_on_dataset(model, single, str(fb), lossm, add_loss, task="19eval")
    test_on_dataset(model, batched, str(fc), lossm, add_loss, task="LA")
    la = fa.read_text().splitlines()
    lb = fb.read_text().splitlines()
    lc = fc.read_text().splitlines()
    assert len(la) == len(lb) == len(lc) == n
    # oracle: eval-mode forward + the same score definitions
    sd = fill_state(o_resnet.resnet18_shapes())
    x = feats.transpose(2, 3).contiguous()
    with torch.no_grad():
        ft, mu = o_resnet.resnet18_forward(sd, x, training=False, noise=None)
        if add_loss is None:
            want = torch.softmax(mu, dim=1)[:, 0]
        else:
            want = -ocsoftmax_forward(ft, fill_value("center", (1, 256)), torch.zeros(n, dtype=torch.long), 0.9, 0.2, 20.0)[1]
    for i in range(n):
        name_a, val_a, key_a = la[i].split(" ")
        name_b, val_b, key_b = lb[i].split(" ")
        assert name_a == name_b == names[i] and lc[i].split(" ")[0] == names[i]
        assert key_a == "bonafide"  # generate_score.py:96 overwrites the labels with zeros
        assert len(lc[i].split(" ")) == 2
        np.testing.assert_allclose(float(val_a), float(val_b), atol=2e-5)   # batch 3 vs batch 1
        np.testing.assert_allclose(float(val_a), float(want[i]), atol=2e-4)  # vs the oracle
        assert float(lc[i].split(" ")[1]) == float(val_a)
    fd = tmp_path / "d.txt"
    test_on_dataset(model, batched, str(fd), lossm, add_loss, task="19eval", keep_dataset_labels=True)
    keys = [ln.split(" ")[2] for ln in fd.read_text().splitlines()]
    assert keys == ["spoof" if int(l) else "bonafide" for l in labels]


@pytest.mark.gpu
def test_score_pcm_equals_trainer_score():
    from asvspoof2021_air_amd.generate_score import score_pcm
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    from oracle.filler import synth_pcm
    model = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(model)
    model.set_attention_noise(None)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    tr = Trainer(model, loss_module=lossm, feat_len=96)
    pcm = synth_pcm(4, 8000, seed=3).cuda()
    a = score_pcm(tr.model, tr.loss, pcm, feat_len=96)
    b = tr.score(pcm)
    assert torch.allclose(a, b, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["resnet", "ecapa"])
def test_graphed_batch1_scoring_equals_eager(tmp_path, kind):
    """The reference scores with batch size 1 (generate_score.py:73): the captured-hipGraph path writes the
    same file as the eager path, bit for bit, and is never slower than launching the kernels one by one
    (measured on MI355X: ResNet-18 is GPU-bound even at batch 1 - 587 utt/s either way - so the graph pays
    only where the launches dominate)."""
    import time
    from asvspoof2021_air_amd.generate_score import GraphedScorer, batch_scores, test_on_dataset
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    if kind == "resnet":
        from asvspoof2021_air_amd.resnet import ResNet
        model = ResNet(3, 256, resnet_type="18", nclasses=2)
        fill_module_(model)
        model.set_attention_noise(None)
    else:
        from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
        model = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        fill_module_(model)
    model = model.cuda().eval()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    n, T = 5, 750
    feats, names, labels = _items(n, T, True)
    tags = torch.zeros(n)
    single = [(feats[i:i + 1], names[i:i + 1], tags[i:i + 1], labels[i:i + 1]) for i in range(n)]
    fa, fb = tmp_path / "eager.txt", tmp_path / "graph.txt"
    test_on_dataset(model, single, str(fa), lossm, "ocsoftmax", task="19eval", ecapa=(kind == "ecapa"))
    test_on_dataset(model, single, str(fb), lossm, "ocsoftmax", task="19eval", ecapa=(kind == "ecapa"), use_graph=True)
    assert fa.read_text() == fb.read_text()
    x = feats[:1].cuda().transpose(2, 3)
    x = (x.squeeze(1) if kind == "ecapa" else x).contiguous()
    gs = GraphedScorer(model, x, lossm, "ocsoftmax")

    def rate(fn, reps=30):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return reps / (time.perf_counter() - t0)

    with torch.no_grad():
        r_eager = rate(lambda: batch_scores(model, x, lossm, "ocsoftmax"))
    r_graph = rate(lambda: gs(x))
    print("%s batch-1 scoring at T=750: eager %.0f utt/s, hipGraph replay %.0f utt/s" % (kind, r_eager, r_graph))
    assert r_graph > 0.9 * r_eager


@pytest.mark.gpu
def test_trainer_checkpoints_feed_generate_score(tmp_path):
    """Trainer.save_checkpoint() writes the reference's files (main_train.py:675-704); loading them the way
    generate_score.py:46-48 does (torch.load of whole-module pickles) scores exactly like the live modules."""
    from asvspoof2021_air_amd.generate_score import test_on_dataset
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    torch.manual_seed(688)
    tr = Trainer(ResNet(3, 256, resnet_type="18", nclasses=2), feat_len=96).set_out_fold(str(tmp_path / "run"))
    feats, names, labels = _items(6, 96, True)
    x = feats.transpose(2, 3).contiguous().cuda()
    for step in range(2):  # two real optimisation steps so the checkpoint is not the initial state
        loss, _ = tr.step_features(x, labels.cuda())
        tr.log_step(0, step, loss)
    assert tr.save_checkpoint(0, val_loss=float(loss)) is True
    lines = (tmp_path / "run" / "train_loss.log").read_text().splitlines()
    assert len(lines) == 2 and lines[1].split("\t")[:2] == ["0", "1"] and float(lines[1].split("\t")[2]) == float(loss)
    feat_model = torch.load(tmp_path / "run" / "anti-spoofing_feat_model.pt", weights_only=False).cuda()
    loss_model = torch.load(tmp_path / "run" / "anti-spoofing_loss_model.pt", weights_only=False).cuda()
    per_epoch = torch.load(tmp_path / "run" / "checkpoint" / "anti-spoofing_feat_model_1.pt", weights_only=False)
    assert list(per_epoch.state_dict().keys()) == list(tr.model.state_dict().keys())
    for (k, a), b in zip(feat_model.state_dict().items(), tr.model.state_dict().values()):
        assert torch.equal(a.cpu(), b.cpu()), k
    tags = torch.zeros(6)
    batches = [(feats[i:i + 3], names[i:i + 3], tags[i:i + 3], labels[i:i + 3]) for i in range(0, 6, 3)]
    tr.model.set_attention_noise(None)
    feat_model.set_attention_noise(None)
    fa, fb = tmp_path / "live.txt", tmp_path / "loaded.txt"
    test_on_dataset(tr.model, batches, str(fa), tr.loss, "ocsoftmax", task="19eval")
    test_on_dataset(feat_model, batches, str(fb), loss_model, "ocsoftmax", task="19eval")
    assert fa.read_text() == fb.read_text() and len(fa.read_text().splitlines()) == 6
