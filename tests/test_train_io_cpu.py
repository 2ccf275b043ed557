"""CPU: text formats of the training logs, checkpoint file names, LR schedule, crop-offset draw, seeded
construction vs the real reference (tests/golden/init_checksums.npz) and the EER function vs its goldens.
None of this touches the GPU (the drop-in modules are constructed, never run)."""
import os

import numpy as np
import pytest
import torch


def test_seeded_construction_equals_reference(golden):
    """torch.manual_seed(688) + construction consumes the RNG exactly like the reference's modules (same
    layer order, including the discarded downsample of resnet.py:162-166): every state_dict tensor has the
    reference's key, order, size, sum, |.|-sum and first element (checksums of the REAL reference,
    tests/golden/make_golden_eer2.py init)."""
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    g = golden("init_checksums.npz")
    for which in ("resnet", "ecapa"):
        torch.manual_seed(int(g["seed"]))
        if which == "resnet":
            net = ResNet(3, 256, resnet_type="18", nclasses=2)
        else:
            net = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        lossmod = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        sd = net.state_dict()
        assert list(sd.keys()) == [str(k) for k in g[which + "_names"]]
        for (k, v), want, bits in zip(sd.items(), g[which + "_vals"], g[which + "_bits"]):
            raw = v.detach().reshape(-1)
            v = raw.double()
            got = [float(v.sum()), float(v.abs().sum()), float(v[0]) if v.numel() else 0.0, float(v.numel())]
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, err_msg="%s %s" % (which, k))
            assert got[2:] == list(want[2:])
            # bit-exact and independent of the summation order: int64 sum of the raw 32-bit patterns
            b = int(raw.contiguous().view(torch.int32).long().sum()) if raw.dtype == torch.float32 else int(raw.long().sum())
            assert b == int(bits), (which, k)
        np.testing.assert_array_equal(lossmod.center.detach().numpy(), g[which + "_center"])


def test_train_log_and_checkpoint_text(tmp_path):
    """train_loss.log / test_loss.log lines and checkpoint file names are the reference's, character for
    character (main_train.py:470-481, :666-667, :675-704)."""
    from asvspoof2021_air_amd import train as T
    tr = T.Trainer.__new__(T.Trainer)  # the writers need no GPU state
    tr.out_fold, tr.world, tr.prev_loss, tr.early_stop_cnt = None, 1, 1e8, 0
    tr.set_out_fold(str(tmp_path))
    assert os.path.isdir(tmp_path / "checkpoint")
    losses = [0.6931471824645996, 3.0517578125e-05, 12.5]
    for i, l in enumerate(losses):
        tr.log_step(0, i, torch.tensor(l))
    tr.log_step(3, 7, 0.25, adv=(1.5, 50.0, 33.33333333333333))
    want = ""
    for i, l in enumerate(losses):
        item = torch.tensor(l).item()
        want += str(0) + "\t" + str(i) + "\t" + str(item) + "\n"  # main_train.py:479-481
    want += str(3) + "\t" + str(7) + "\t" + str(1.5) + "\t" + str(50.0) + "\t" + str(33.33333333333333) + "\t" + str(0.25) + "\n"
    assert (tmp_path / "train_loss.log").read_text() == want
    tr.log_eval(2, [0.5, float("nan"), 0.25], 0.0123)
    assert (tmp_path / "test_loss.log").read_text() == str(2) + "\t" + str(np.nanmean([0.5, float("nan"), 0.25])) + "\t" + str(0.0123) + "\n"
    # checkpoints: whole-module pickles under the reference's names
    tr.model, tr.loss = torch.nn.Linear(2, 2), torch.nn.Linear(2, 1)
    assert tr.save_checkpoint(0, val_loss=0.7) is True
    assert tr.save_checkpoint(1, val_loss=0.9) is False and tr.early_stop_cnt == 1
    assert tr.save_checkpoint(2, val_loss=0.6) is True and tr.early_stop_cnt == 0
    names = sorted(os.listdir(tmp_path / "checkpoint"))
    assert names == sorted(["anti-spoofing_feat_model_%d.pt" % k for k in (1, 2, 3)] +
                           ["anti-spoofing_loss_model_%d.pt" % k for k in (1, 2, 3)])
    assert os.path.exists(tmp_path / "anti-spoofing_feat_model.pt") and os.path.exists(tmp_path / "anti-spoofing_loss_model.pt")
    m = torch.load(tmp_path / "anti-spoofing_feat_model.pt", weights_only=False)  # generate_score.py:46
    assert isinstance(m, torch.nn.Linear)
    tr.set_out_fold(str(tmp_path))  # a fresh run starts its logs over (main_train.py:113)
    assert not os.path.exists(tmp_path / "train_loss.log")


def test_step_lr_schedule():
    """lr = lr0 * decay^(epoch // interval) on every optimiser (main_train.py:144-147, :294-298)."""
    from asvspoof2021_air_amd import train as T
    from asvspoof2021_air_amd.optim import FusedAdam, FusedSGD
    tr = T.Trainer.__new__(T.Trainer)
    tr.lr0 = 5e-4
    tr.feat_optimizer = FusedAdam(torch.nn.Linear(2, 2), lr=5e-4)
    tr.loss_optimizer = FusedSGD(torch.nn.Linear(2, 1), lr=5e-4)
    for epoch, want in ((0, 5e-4), (29, 5e-4), (30, 2.5e-4), (59, 2.5e-4), (60, 1.25e-4), (90, 6.25e-5)):
        tr.set_epoch(epoch)
        assert tr.feat_optimizer.param_groups[0]["lr"] == want and tr.loss_optimizer.param_groups[0]["lr"] == want
    tr.set_epoch(7, lr_decay=0.5, interval=4)  # --interval 4 (the EER goldens' schedule)
    assert tr.feat_optimizer.param_groups[0]["lr"] == 2.5e-4
    assert T.adjust_learning_rate(1e-4, tr.loss_optimizer, 61) == 1e-4 * 0.25


def test_chop_starts_draw_like_reference(golden):
    """np.random.randint(T - feat_len): exclusive upper bound, one draw per item in order (dataset.py:69);
    the golden starts are the reference's own draws."""
    from asvspoof2021_air_amd.dataset import chop_starts
    from oracle import pad as o_pad
    g = golden("pad.npz")
    assert chop_starts(750, 750, 4) is None and chop_starts(401, 750, 4) is None
    np.random.seed(0)
    got = chop_starts(900, 750, 6)
    np.random.seed(0)
    want = [np.random.randint(150) for _ in range(6)]
    assert got.dtype == torch.int32 and got.tolist() == want and max(want) < 150
    rng = np.random.RandomState(3)
    assert chop_starts(751, 750, 50, rng).tolist() == [0] * 50  # T - feat_len = 1: only offset 0 is ever drawn
    # and the oracle's pad_chop (pinned to the reference by pad.npz) crops at exactly these offsets
    spec = torch.arange(900, dtype=torch.float32).view(1, 900, 1).repeat(1, 1, 2)
    np.random.seed(0)
    rows = [int(o_pad.pad_chop(spec, 750)[0, 0, 0]) for _ in range(6)]
    assert rows == want


def test_eval_metrics_matches_goldens(golden):
    from asvspoof2021_air_amd.eval_metrics import compute_det_curve, compute_eer, eer_both_polarities
    from oracle import eer as o_eer
    g = golden("eer.npz")
    for a, b, k in (("tgt", "non", 0), ("tgt_t", "non_t", 1)):  # the second pair carries tied scores
        e, t = compute_eer(g[a], g[b])
        np.testing.assert_allclose([e, t], [g["eer"][k], g["thr"][k]], atol=1e-12)
        for got, want in zip(compute_det_curve(g[a], g[b]), o_eer.det_curve(g[a], g[b])):
            np.testing.assert_array_equal(got, want)
    rng = np.random.default_rng(0)
    for _ in range(100):  # heavy ties, tiny sets
        t = np.round(rng.standard_normal(rng.integers(1, 30)), 1)
        n = np.round(rng.standard_normal(rng.integers(1, 30)) + 0.5, 1)
        assert compute_eer(t, n) == o_eer.compute_eer(t, n)
    s = np.r_[g["tgt"], g["non"]]
    lab = np.r_[np.zeros(len(g["tgt"])), np.ones(len(g["non"]))]
    assert eer_both_polarities(s, lab) == min(compute_eer(g["tgt"], g["non"])[0], compute_eer(-g["tgt"], -g["non"])[0])


def test_pad_mode_names():
    from asvspoof2021_air_amd.feature_extraction import pad_mode_id
    assert [pad_mode_id(p) for p in ("repeat", "zero", "silence")] == [0, 1, 2]
    with pytest.raises(ValueError):
        pad_mode_id("edge")
