"""GPU: EER parity in a regime where the reference separates the classes (round 3 fixtures).

tests/golden/synth_eer3_{resnet,ecapa}.npz hold the REAL reference (its LFCC, ResNet-18 / ECAPA-TDNN-512,
AngularIsoLoss, torch.optim Adam + SGD, its own step decay with --interval 4) trained from the seeded
construction (torch.manual_seed(688)) for 16 epochs on the separable synthetic corpus
(asvspoof2021_air_amd/synth.py, mix_lo = 0.4; 768 training utterances of 1 s, batch 32) and scored on 4096
HELD-OUT utterances: 13 wrong trials of 4096 (EER 0.34 % ResNet, 0.29 % ECAPA).  synth_eer4s_* is BASELINE's
workload shape: 4 s utterances, 401 LFCC frames repeat-padded to feat_len 750, batch 64, 12 epochs, 1024 held-out.
The HIP path runs the same recipe from raw PCM (fused LFCC -> model -> OC-Softmax -> Adam + SGD) from the same
seeded construction - bit-equal to the reference's (tests/test_train_io_cpu.py) - and must land within THREE
TRIALS of the reference's EER; both systems' error counts at their EER thresholds are printed.

How tight the LOSS CURVE can be is set by the training run itself, not by the kernels: it is chaotic from the
third optimisation step on (Adam's first updates are lr * sign(g), so rounding-level gradient differences flip
whole updates; tools/dbg_eer_epoch0.py).  The curve is therefore held by its envelope (below); the EER, which is
a property of where the run converges, is held tightly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_CORPUS = {}


def _corpus(g):
    from asvspoof2021_air_amd.synth import corpus
    L, B, NTR, NHO, EPOCHS, INTERVAL, FL = [int(v) for v in g["cfg"]]
    key = (L, NTR, NHO, float(g["mix_lo"]))
    if key not in _CORPUS:  # ~25 s of host numpy: shared by the three tests of this module
        pcm_tr, lab_tr = corpus(688, NTR, L, mix_lo=float(g["mix_lo"]))
        pcm_ho, lab_ho = corpus(689, NHO, L, mix_lo=float(g["mix_lo"]))
        _CORPUS[key] = (pcm_tr, lab_tr, pcm_ho, lab_ho)
    pcm_tr, lab_tr, pcm_ho, lab_ho = _CORPUS[key]
    np.testing.assert_allclose([pcm_tr.astype(np.float64).sum(), pcm_ho.astype(np.float64).sum()], g["pcm_sum"], rtol=1e-9)
    np.testing.assert_array_equal(lab_ho, g["labels"])
    return pcm_tr, lab_tr, pcm_ho, lab_ho


def _run(g, which, dtype, perturb=0):
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.eval_metrics import eer_both_polarities
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    L, B, NTR, NHO, EPOCHS, INTERVAL, FL = [int(v) for v in g["cfg"]]
    pcm_tr, lab_tr, pcm_ho, lab_ho = _corpus(g)
    torch.manual_seed(int(g["seed"]))  # the reference's construction order: model, then the loss centre
    if which == "resnet":
        model = ResNet(3, 256, resnet_type="18", nclasses=2)
    else:
        model = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        model.set_compute_dtype(dtype)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    if perturb:  # tools/dbg_eer_spread.py: the HIP path's own run-to-run spread (one weight moved by 1e-7)
        with torch.no_grad():
            model.conv1.weight.view(-1)[perturb] *= (1.0 + 1e-7)
            if dtype == "bf16":  # conv1's weight is a bf16 operand there (1e-7 rounds away): move an fp32-read one too
                model.bn1.weight.view(-1)[perturb] *= (1.0 + 1e-7)
    tr = Trainer(model, loss_module=lossm, feat_len=FL, ecapa=(which == "ecapa"))  # T < FL: repeat-padded
    TA = FL
    for _ in range(3):
        TA = (TA + 2 - 3) // 2 + 1
    xtr, ltr = torch.from_numpy(pcm_tr).cuda(), torch.from_numpy(lab_tr).cuda()
    xho = torch.from_numpy(pcm_ho).cuda()

    def noise(seed):  # replay the reference's host-side attention noise (resnet.py:38-42)
        if which == "resnet":
            torch.manual_seed(seed)
            model.set_attention_noise(1e-5 * torch.randn(B, TA, 256))

    epoch_loss, step = [], 0
    for ep in range(EPOCHS):
        tr.set_epoch(ep, lr_decay=0.5, interval=INTERVAL)  # main_train.py:294-298
        tot = 0.0
        for i in range(0, NTR, B):
            noise(9000 + step)
            loss, _ = tr.step(xtr[i:i + B], ltr[i:i + B])
            tot += loss.item()
            step += 1
        epoch_loss.append(tot / (NTR // B))
    scores = []
    for i in range(0, NHO, B):
        noise(9500 + i // B)
        scores.append(tr.score(xho[i:i + B]).cpu())
    scores = torch.cat(scores).numpy()
    eer = eer_both_polarities(scores, lab_ho)
    return tr, model, lossm, np.array(epoch_loss), scores, eer, lab_ho, pcm_ho


def _eer_first(scores, labels):
    from asvspoof2021_air_amd.eval_metrics import eer_both_polarities
    return eer_both_polarities(scores, labels)


def _error_counts(scores, labels):
    from asvspoof2021_air_amd.eval_metrics import compute_eer
    eer, thr = compute_eer(scores[labels == 0], scores[labels == 1])
    return int((scores[labels == 0] < thr).sum()), int((scores[labels == 1] >= thr).sum())


def _spread(golden, fixture):
    """The reference's OWN run-to-run spread (make_golden_eer3.py ``spread``: the same recipe from initial weights
    moved by 1e-7 in one element - the trajectory is chaotic, so this is a second / third sample of it)."""
    import os
    from conftest import GOLDEN
    path = fixture.replace(".npz", "_spread.npz")
    return golden(path) if os.path.exists(os.path.join(GOLDEN, path)) else None


class _Samples(list):
    """EERs of further HIP runs (a list of floats, as before) that also carries their final-epoch losses and curves."""
    final_loss = ()
    epoch_loss = ()


def _more_eers(g, which, dtype, n=2):
    """EERs of n more HIP runs of the same recipe, each from initial weights with one element moved by 1e-7
    (``.final_loss``: their last-epoch mean losses - the convergence floor is a sample of the same chaos)."""
    runs = [_run(g, which, dtype, perturb=k) for k in range(1, n + 1)]
    out = _Samples(float(r[5]) for r in runs)
    out.final_loss = tuple(float(r[3][-1]) for r in runs)
    out.epoch_loss = tuple(np.asarray(r[3], dtype=np.float64) for r in runs)
    return out


def chaos_gate(fixture="eer_chaos_4s_resnet.json"):
    """Thresholds for the final-epoch loss of a SAMPLE of runs, from the committed empirical distribution of 2 x 20 GPU
    runs (tests/golden/eer_chaos_4s_resnet.json; VERDICT r5 item 8: the evidence was a markdown table): `tail` = its
    92.5th percentile - at most one of three samples may lie above it (false alarm 3 p^2 (1 - p) + p^3 = 1.6 % at
    p = 3 / 40, stated by tests/test_eer_fixture_cpu.py) -, `cap` = 1.5 x its maximum - none above -, `median_hi` = its
    90th percentile for the median of the samples."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, fixture)) as f:
        d = json.load(f)
    allv = np.array([v for arm in d["final_loss"].values() for v in arm], dtype=np.float64)
    return {"tail": float(np.quantile(allv, 0.925)), "cap": 1.5 * float(allv.max()),
            "median_hi": float(np.quantile(allv, 0.90)), "n": int(allv.size), "values": allv,
            "arms": {k: np.array(v, dtype=np.float64) for k, v in d["final_loss"].items()},
            "reference": float(d["reference_final_loss"])}


def _check(g, epoch_loss, scores, eer, lab_ho, name, curve_rtol, spread=None, more=(), chaos=None):
    from _budget import record
    ref_eer = float(g["eer"])
    n_side = int(min((lab_ho == 0).sum(), (lab_ho == 1).sum()))
    mine, ref = _error_counts(scores, lab_ho), [int(v) for v in g["errors"]]
    print("%s epoch losses %s\nreference        %s\nEER %.5f vs reference %.5f; wrong trials (bona fide, spoofed) %s vs "
          "reference %s of %d per side" % (name, np.round(epoch_loss, 4).tolist(), np.round(g["epoch_loss"], 4).tolist(),
                                           eer, ref_eer, mine, ref, n_side))
    record("eer[%s]" % name, {"eer": float(eer), "ref_eer": ref_eer, "errors": list(mine), "ref_errors": ref,
                               "final_loss": float(epoch_loss[-1]), "ref_final_loss": float(g["epoch_loss"][-1])})
    assert ref_eer < 0.05                       # the regime: the reference separates the classes
    # The held-out EER of ONE run is a sample of a chaotic trajectory: the reference's own runs from initial weights
    # with a single element moved by 1e-7 make 13 / 27 / 25 wrong trials (ResNet) and 13 / 7 / 3 (ECAPA), this path's
    # 31 / 19 / 15 / 1 / 7 (ECAPA fp32, tools/dbg_eer_spread.py).  So samples are compared with samples: the MEDIAN
    # of this path's runs (the unperturbed one + `more`) must lie inside the range of the reference's runs widened
    # by three trials on either side.  Without spread fixtures: within three trials of the single reference run.
    ref_all = [ref_eer] + ([float(v) for v in spread["eer"]] if spread is not None else [])
    mine_all = [float(eer)] + [float(v) for v in more]
    med = float(np.median(mine_all))
    print("EER samples: this path %s (median %.5f), reference %s" % (np.round(mine_all, 5).tolist(), med,
                                                                     np.round(ref_all, 5).tolist()))
    record("eer_samples[%s]" % name, {"hip": mine_all, "reference": ref_all})
    tri = 3.0 / n_side
    assert min(ref_all) - tri - 1e-12 <= med <= max(ref_all) + tri + 1e-12, (mine_all, ref_all)
    # final-epoch loss: the converged floor.  One run's last epoch can sit on a transient (round 4: a build that
    # differed only in the summation order of the BatchNorm statistics ended one 4 s run at 0.139 where three others
    # and the reference's own perturbed runs end at 0.078 - 0.085), so where further samples of this path exist the
    # MEDIAN of their floors is what must match the reference's
    floors = [float(epoch_loss[-1])] + [float(v) for v in getattr(more, "final_loss", ())]
    np.testing.assert_allclose(np.median(floors), g["epoch_loss"][-1], rtol=0.40)
    # Round 5 (profiles/r05_eer_chaos.md): 20 runs each of the all-f32 and the split-bf16 configuration at the 4 s shape
    # (one initial weight moved by 1e-7) end at a median of 0.0797 / 0.0802 (reference 0.0786) with 2 / 1 of 20 runs
    # still on a transient at the last epoch (0.17, 0.21 / 0.16): a single run above 0.15 is a 5 - 10 % event of EITHER
    # arithmetic, so the gate is on the samples - at most one of them above 0.15, none above 0.30 - and the unperturbed
    # run's last epoch must still be its floor
    if chaos is None:
        assert sum(f >= 0.15 for f in floors) <= (1 if len(floors) >= 3 else 0) and max(floors) < 0.30, floors
    else:
        # (round 6) the same gate with its constants READ from the committed distribution of those 40 runs instead of
        # hand-set: at most one sample above its 92.5th percentile (0.110), none above 1.5 x its maximum (0.31), and
        # the samples' median below its 90th percentile (0.0975; ADVICE r5: an absolute bound on the median)
        assert len(floors) >= 3, "the 'at most one' clause needs three samples"
        assert sum(f > chaos["tail"] for f in floors) <= 1 and max(floors) < chaos["cap"], (floors, chaos["tail"], chaos["cap"])
        assert float(np.median(floors)) <= chaos["median_hi"], (floors, chaos["median_hi"])
    assert epoch_loss[-1] <= 1.05 * epoch_loss[-4:].min()  # ... and it IS a floor
    # first epoch (24 steps, Adam's first updates are lr * sign(g)): builds of this round that differ only in the
    # summation order of one weight-gradient kernel gave 3.57 and 4.1 against the reference's 4.32
    np.testing.assert_allclose(epoch_loss[0], g["epoch_loss"][0], rtol=curve_rtol[0])
    # mid-training the two runs are different samples of a chaotic trajectory: the reference's own curve is not
    # monotone (0.09 -> 0.126 at epoch 11), and on the HIP side a change of the SUMMATION ORDER inside the
    # BatchNorm reductions alone moved a transient spike (1.01 -> 2.28 -> 0.53 around epoch 4) in and out of the
    # run.  What is stable is the envelope: the running minimum of the epoch loss stays within a factor
    # curve_rtol[1] of the reference's, no epoch climbs back above the first one, and both converge to the same
    # place (the final-epoch check above).
    ref_min = np.minimum.accumulate(g["epoch_loss"])
    ratio = np.minimum.accumulate(epoch_loss) / ref_min
    # (slower than the reference by at most curve_rtol[1]; FASTER is bounded more loosely - three builds of this round,
    # differing only in summation orders, gave running-minimum ratios between 0.27 and 2.4)
    hi, lo = curve_rtol[1], 1.0 / (1.5 * curve_rtol[1])
    worst, best = float(ratio.max()), float(ratio.min())
    if spread is not None and "epoch_loss" in getattr(spread, "files", ()):
        # The yardstick where it exists (round 4: the spread fixture carries whole curves): the REFERENCE'S OWN perturbed
        # runs against its unperturbed curve lag it by up to 2.3x and lead it by up to 3x.  Samples are compared with
        # samples here too: the MEDIAN over this path's runs (the unperturbed one + `more`) of the worst lag / lead
        # must stay inside the reference's own range + 25 % (a build of round 4 that only changed the tap order inside
        # the stride-2 data gradient sat one run on a transient at epoch 8 - 3.8x - while its perturbed runs, its
        # final loss, its EER and every other check agreed with the reference).
        own = np.array([np.minimum.accumulate(c) / ref_min for c in spread["epoch_loss"]])
        mine = [ratio] + [np.minimum.accumulate(c) / ref_min for c in getattr(more, "epoch_loss", ())]
        worst = float(np.median([r.max() for r in mine]))
        best = float(np.median([r.min() for r in mine]))
        print("running-minimum ratio: this path's runs max %s min %s; the reference's own runs max %s min %s" % (
            np.round([r.max() for r in mine], 2).tolist(), np.round([r.min() for r in mine], 2).tolist(),
            np.round(own.max(1), 2).tolist(), np.round(own.min(1), 2).tolist()))
        record("curve_envelope[%s]" % name, {"hip_max": [float(r.max()) for r in mine], "ref_max": own.max(1).tolist()})
        hi, lo = max(hi, 1.25 * float(own.max())), min(lo, float(own.min()) / 1.25)
    assert worst <= hi and best >= lo, (ratio, worst, best, hi, lo)
    assert epoch_loss[1:].max() <= epoch_loss[0], epoch_loss
    # the two systems rank the held-out set alike: the reference's threshold-free separation carries over
    bona, spoof = scores[lab_ho == 0], scores[lab_ho == 1]
    assert np.mean(bona) - np.mean(spoof) > 0.5 and np.mean(g["scores"][lab_ho == 0]) - np.mean(g["scores"][lab_ho == 1]) > 0.5


def test_synthetic_corpus_eer_matches_reference(golden):
    g = golden("synth_eer3_resnet.npz")
    tr, model, lossm, epoch_loss, scores, eer, lab_ho, pcm_ho = _run(g, "resnet", "fp32")
    _check(g, epoch_loss, scores, eer, lab_ho, "resnet", (0.25, 3.5), _spread(golden, "synth_eer3_resnet.npz"),
           _more_eers(g, "resnet", "fp32"))
    NO = 512  # the oracle re-scores the first 512 held-out utterances (CPU time)
    scores, lab_ho, pcm_ho = scores[:NO], lab_ho[:NO], pcm_ho[:NO]
    eer = _eer_first(scores, lab_ho)
    # score parity proper: the ORACLE scores the held-out set with the weights the HIP path trained.
    # Same weights -> same scores (1e-3) -> same EER (to one trial).
    from oracle import eer as o_eer, lfcc as o_lfcc, resnet as o_resnet
    from oracle.loss import ocsoftmax_forward
    B = int(g["cfg"][1])
    TA = 1 + int(g["cfg"][0]) // 160
    for _ in range(3):
        TA = (TA + 2 - 3) // 2 + 1
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    centre = lossm.center.detach().cpu()
    xo = torch.from_numpy(o_lfcc.lfcc_forward(pcm_ho.copy())).unsqueeze(1).transpose(2, 3).contiguous()
    o_scores = []
    with torch.no_grad():
        for i in range(0, len(lab_ho), B):
            torch.manual_seed(9500 + i // B)
            nz = 1e-5 * torch.randn(B, TA, 256)
            ft, _ = o_resnet.resnet18_forward(sd, xo[i:i + B], training=False, noise=nz)
            o_scores.append(-ocsoftmax_forward(ft, centre, torch.zeros(B, dtype=torch.long), 0.9, 0.2, 20.0)[1])
    o_scores = torch.cat(o_scores).numpy()
    np.testing.assert_allclose(scores, o_scores, atol=1e-3)
    assert abs(eer - o_eer.eer_both_polarities(o_scores, lab_ho)) <= 1.0 / 256


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_synthetic_corpus_eer_ecapa(golden, dtype):
    """ECAPA-TDNN-512 in the reference's fp32 arithmetic and with bf16-resident activations (BASELINE configs[2],
    compute_dtype "bf16") against the same fp32 reference run."""
    g = golden("synth_eer3_ecapa.npz")
    tr, model, lossm, epoch_loss, scores, eer, lab_ho, pcm_ho = _run(g, "ecapa", dtype)
    _check(g, epoch_loss, scores, eer, lab_ho, "ecapa " + dtype, (0.25, 3.5), _spread(golden, "synth_eer3_ecapa.npz"),
           _more_eers(g, "ecapa", dtype, 4))
    NO = 512
    scores, lab_ho, pcm_ho = scores[:NO], lab_ho[:NO], pcm_ho[:NO]
    eer = _eer_first(scores, lab_ho)
    from oracle import ecapa as o_ecapa, eer as o_eer, lfcc as o_lfcc
    from oracle.loss import ocsoftmax_forward
    B = int(g["cfg"][1])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    centre = lossm.center.detach().cpu()
    xo = torch.from_numpy(o_lfcc.lfcc_forward(pcm_ho.copy())).transpose(1, 2).contiguous()
    o_scores = []
    with torch.no_grad():
        for i in range(0, len(lab_ho), B):
            ft, _ = o_ecapa.ecapa_forward(sd, xo[i:i + B], training=False, bf16=("resident" if dtype == "bf16" else False))
            o_scores.append(-ocsoftmax_forward(ft, centre, torch.zeros(B, dtype=torch.long), 0.9, 0.2, 20.0)[1])
    o_scores = torch.cat(o_scores).numpy()
    np.testing.assert_allclose(scores, o_scores, atol=1e-3 if dtype == "fp32" else 1e-2)
    assert abs(eer - o_eer.eer_both_polarities(o_scores, lab_ho)) <= (1.0 / 256 if dtype == "fp32" else 4.0 / 256)


@pytest.mark.parametrize("which,dtype", [("resnet", "fp32"), ("ecapa", "bf16")])
def test_synthetic_corpus_eer_at_baseline_shape(golden, which, dtype):
    """BASELINE's workload shape: 4 s utterances, 401 LFCC frames repeat-padded to feat_len 750
    inside the fused front-end, batch 64, 12 epochs with the reference's step decay, 1024 held-out utterances -
    against the real reference trained the same way (synth_eer4s_*.npz)."""
    g = golden("synth_eer4s_%s.npz" % which)
    tr, model, lossm, epoch_loss, scores, eer, lab_ho, pcm_ho = _run(g, which, dtype)
    # (round 4: this path is sampled three times at this shape - the unperturbed run and two from initial weights with
    # one element moved by 1e-7 - and its MEDIAN is what is compared; the reference's own spread exists for both
    # models: synth_eer4s_{resnet,ecapa}_spread.npz.  A single bf16 ECAPA run of a build that only regrouped the
    # BatchNorm statistics' partial sums made 9 wrong trials of 1010 where the reference's single run makes 3.)
    spread = _spread(golden, "synth_eer4s_%s.npz" % which)
    more = _more_eers(g, which, dtype, 2)
    _check(g, epoch_loss, scores, eer, lab_ho, "%s %s 4 s" % (which, dtype), (0.25, 3.5), spread, more,
           chaos=chaos_gate() if which == "resnet" else None)
