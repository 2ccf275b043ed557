"""GPU: the hot path at BASELINE.json's FULL sizes (configs[1]: ResNet-18 fp32, batch 64, 4 s @ 16 kHz,
feat_len 750; configs[2]: ECAPA-TDNN-512, T = 750) - one whole train step against the CPU oracle, plus
size-independent properties (determinism, gradient linearity in the loss weight, batch independence in
eval mode).  The oracle needs ~10-30 s of CPU per case at these sizes."""
import numpy as np
import pytest
import torch

from oracle import ecapa as o_ecapa
from oracle import lfcc as o_lfcc, pad as o_pad, resnet as o_resnet, train as o_train
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat, synth_pcm

from _budget import check_bf16_band, check_relu_flips, conv_path, record, tol  # noqa: E402

pytestmark = pytest.mark.gpu


def _resnet_trainer(feat_len=750):
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    m = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(m)
    m.set_attention_noise(None)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    return Trainer(m, loss_module=lossm, feat_len=feat_len)


_FULL = {}


def _full_size_oracle():
    """The CPU side of the full-size step (fp32 oracle step + fp64 gradients), computed once for both paths."""
    if not _FULL:
        B, L, FL = 64, 64000, 750
        pcm = synth_pcm(B, L, seed=688)
        g = torch.Generator().manual_seed(1)
        labels = (torch.rand(B, generator=g) < 0.9).long()
        labels[0], labels[1] = 0, 1
        feat = torch.from_numpy(o_lfcc.lfcc_forward(pcm.numpy().copy()))
        assert feat.shape == (B, 401, 60)
        xin = torch.stack([o_pad.repeat_pad(feat[b:b + 1], FL) for b in range(B)])
        xo = o_pad.to_model_input(xin).contiguous()
        otr = o_train.OracleTrainer("resnet", fill_state(o_resnet.resnet18_shapes()), fill_value("center", (1, 256)))
        lo, no, _, go, _ = otr.step(xo, labels, None)
        p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in fill_state(o_resnet.resnet18_shapes()).items()}
        o64 = o_train.OracleTrainer("resnet", p64, fill_value("center", (1, 256)).double())
        _, _, _, g64, _, _ = o64.loss_and_grads(xo.double(), labels, None)
        _FULL.update(pcm=pcm, labels=labels, lo=lo, no=no, go=go, g64=g64, otr=otr, xo=xo)
    return _FULL


@pytest.mark.parametrize("path", ["strict", "default"])
def test_resnet_full_size_step_vs_oracle(path):
    """BASELINE configs[1] exactly: 64 x 64000 PCM -> LFCC (401 frames) -> repeat-pad 750 -> ResNet-18 ->
    OC-Softmax -> backward -> Adam + SGD, against the oracle on the same inputs.  ``strict`` = round 1's kernels
    (NO_WINO4 = 1: F(2x2,3x3) + direct) with the round-1 slack (1e-3); ``default`` = the Winograd kernels with that
    slack times the emulated per-convolution rounding ratio (tests/_budget.py)."""
    B, L, FL = 64, 64000, 750
    o = _full_size_oracle()
    pcm, labels, lo, no, go, g64, otr = (o[k] for k in ("pcm", "labels", "lo", "no", "go", "g64", "otr"))
    with conv_path(path):
        tr = _resnet_trainer(FL)
        tr.model.keep_saved_for_test = True  # (keeps the tensors saved for backward: the ReLU decisions, below)
        loss, neg = tr.step(pcm.cuda(), labels.cuda())
        torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in tr.model.named_parameters() if p.grad is not None}
    np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-4)
    np.testing.assert_allclose(neg.cpu().numpy(), no.numpy(), atol=2e-4)
    # Gradients.  At this size (55 M activations per layer) some pre-activations sit within fp32 rounding
    # of a ReLU threshold, and a flipped unit moves whole gradient elements: the fp32 CPU oracle ITSELF is
    # up to 5e-2 of max away from its fp64 evaluation on single elements (layer4.1.conv1.weight), and so is
    # the HIP path.  So: relative L2 per tensor against the fp64 oracle, bounded by what the fp32 oracle
    # shows against the same fp64 truth (x3) plus a slack: 1e-3 for the direct kernels (round 1's constant);
    # the Winograd convolutions round at a larger multiple of a layer's output scale and their slack is 1e-3
    # times the emulated rounding ratio of the two kernels (tests/_budget.py: about 5e-3; measured 3.6e-3 on
    # the worst tensor, layer4.1.bn2.weight).
    slack = tol("full_size_slack", path)
    worst = ("", 0.0, 0.0)
    for k, gh in grads.items():
        ref = g64[k].numpy().ravel()
        nrm = np.linalg.norm(ref) + 1e-30
        e_hip = np.linalg.norm(gh.ravel().astype(np.float64) - ref) / nrm
        e_cpu = np.linalg.norm(go[k].numpy().ravel().astype(np.float64) - ref) / nrm
        # per tensor against the PLAIN fp64 oracle, both paths (the default path kept this bound for its worst
        # tensor only): strict at the slack, default at twice the derived slack - the binding check for default is the
        # masked oracle below, this one keeps a single bad tensor from hiding behind it
        assert e_hip <= 3.0 * e_cpu + (slack if path == "strict" else 2.0 * slack), (path, k, e_hip, e_cpu, slack)
        if e_hip > worst[1]:
            worst = (k, e_hip, e_cpu)
    print("worst relative L2 gradient error vs fp64: %s hip %.2e (fp32 CPU oracle: %.2e)" % worst)
    record("resnet_full_size_worst_grad[%s]" % path, list(worst))
    if path == "default":
        # Round 4.  The distance to the PLAIN fp64 oracle above is dominated by ReLU decisions on pre-activations
        # within rounding of 0 (test_resnet_gpu.py::test_grads_vs_oracle_small proves it at (2, 96): 3 flips of 1.8 M
        # account for a 1000x gap), and which of them flip changes with every change of a summation order - the
        # BatchNorm statistics taken in the convolution epilogues moved layer4.1.bn2.weight from 3.6e-3 to 6.3e-3
        # while agreeing with the BatchNorm's own passes to 2e-6 at kernel level.  So for the Winograd path the
        # bound is on what is computed, not on which side of 0 a rounding error fell: the fp64 oracle evaluated
        # WITH this run's 18 ReLU decisions (oracle/resnet.py::ReluProbe) must agree with every gradient tensor to
        # 5e-4 relative L2 (a tenth of the derived allowance; measured 1.04e-4), the flips are counted and must be few
        # and small, and the plain distance stays bounded at twice the derived allowance.
        S = tr.model._last_saved_for_test
        masks = [S["blocks"][0][1] > 0]
        for blk in S["blocks"]:
            masks += [blk[5] > 0, blk[6] > 0]
        masks.append((S["a5v"] > 0).unsqueeze(2))
        probe = o_resnet.ReluProbe([mk.cpu() for mk in masks])
        p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in fill_state(o_resnet.resnet18_shapes()).items()}
        o64m = o_train.OracleTrainer("resnet", p64, fill_value("center", (1, 256)).double())
        o64m.relu = probe
        xo = o["xo"]
        _, _, _, gm, _, _ = o64m.loss_and_grads(xo.double(), labels, None)
        n_act, n_flip = sum(int(mk.numel()) for mk in masks), sum(probe.flips)
        worst_m = ("", 0.0)
        for k, gh in grads.items():
            ref = gm[k].numpy().ravel()
            e = np.linalg.norm(gh.ravel().astype(np.float64) - ref) / (np.linalg.norm(ref) + 1e-30)
            if e > worst_m[1]:
                worst_m = (k, e)
            assert e <= 5e-4, (k, e)   # measured 1.04e-4 (bn1.weight); the plain distance of the same run is 1.5e-2
        print("ReLU decisions that differ from the fp64 oracle's: %d of %d (largest |pre-activation| %.2e); under this "
              "run's decisions the worst tensor is %s at %.2e" % (n_flip, n_act, max(probe.flip_mag), worst_m[0], worst_m[1]))
        record("resnet_full_size_relu_flips[default]", {"flips": n_flip, "of": n_act, "max_abs_preact": max(probe.flip_mag),
                                                        "masked_worst": list(worst_m)})
        # measured: 557 of 450,289,664 ReLU inputs, the largest |BatchNorm output| among them 4.7e-5.  Round 5: WHICH units
        # may flip is tied to the emulated rounding per ReLU (tests/_budget.py::check_relu_flips: |pre-activation| <=
        # FLIP_K x accumulated emulated conv error x the ReLU input's scale, count <= units x 0.8 x that error) instead of
        # the flat 2500 / 2e-4 of round 4
        rows = check_relu_flips(probe, path, "full size")
        record("resnet_full_size_relu_flip_table[default]", rows)
        assert n_flip <= 1200 and max(probe.flip_mag) <= 1e-4, (n_flip, max(probe.flip_mag))
    # the updated weights (Adam, lr 5e-4: every element moves by ~lr in step 1) agree to a fraction of a step
    w = tr.model.state_dict()["layer4.1.conv2.weight"].cpu().numpy()
    assert np.abs(w - otr.params["layer4.1.conv2.weight"].numpy()).max() <= 2 * 5e-4 + 1e-6


def test_resnet_full_size_properties():
    B, FL = 64, 750
    x = synth_feat((B, 1, 60, FL), seed=3).cuda()
    labels = (torch.arange(B) % 5 != 0).long().cuda()
    # (a) determinism: two identical steps from identical state give bit-identical gradients
    gs = []
    for _ in range(2):
        tr = _resnet_trainer(FL)
        tr.model.train()
        feats, _ = tr.model(x)
        loss, _ = tr.loss(feats, labels)
        loss.backward()
        gs.append(tr.model.arena().grad.clone())
    assert torch.equal(gs[0], gs[1])
    # (b) linearity: backward of 2 * loss doubles every gradient (to rounding)
    tr = _resnet_trainer(FL)
    tr.model.train()
    feats, _ = tr.model(x)
    loss, _ = tr.loss(feats, labels)
    (2.0 * loss).backward()
    g2 = tr.model.arena().grad
    scale = float(gs[0].abs().max())
    assert float((g2 - 2.0 * gs[0]).abs().max()) <= 2e-5 * scale
    # (c) eval mode: utterances are independent - scoring 64 at once equals scoring them 8 at a time
    tr.model.eval()
    with torch.no_grad():
        f_all, _ = tr.model(x)
        f_parts = torch.cat([tr.model(x[i:i + 8])[0] for i in range(0, B, 8)])
    assert float((f_all - f_parts).abs().max()) <= 2e-5 * float(f_all.abs().max())


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16c"])
def test_ecapa_full_length_step_vs_oracle(dtype):
    """ECAPA-TDNN-512 at the reference frame count T = 750 (B = 16 keeps the CPU oracle to seconds):
    loss and every gradient as relative L2 per tensor; fp32 within 5e-3 of the fp64 oracle, bf16 against
    the bf16 oracle inside the band that oracle itself shows between its fp32 and fp64 evaluations
    (oracle/train.py::bf16_gradient_band)."""
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    B, T = 16, 750
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    fill_module_(m)
    m = m.cuda().train().set_compute_dtype(dtype)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    x = synth_feat((B, 60, T), seed=750)
    labels = (torch.arange(B) % 3 != 0).long()
    feat, _ = m(x.cuda())
    loss, _ = lossm(feat, labels.cuda())
    loss.backward()
    got = {k: p.grad.cpu().double().numpy().ravel() for k, p in m.named_parameters() if p.grad is not None}
    if dtype != "fp32":  # "bf16" = resident activations, "bf16c" = bf16 compute on fp32 tensors (oracle/ecapa.py)
        band, errs = o_train.bf16_gradient_band(x, labels, got, "resident" if dtype == "bf16" else True)
        np.testing.assert_allclose(loss.item(), band["loss64"], rtol=2e-3)
        check_bf16_band(errs, band)
        print("%s worst relative L2 %.3g (oracle's own fp32-vs-fp64 worst %.3g)" % (dtype, max(e for e, _ in errs.values()), band["max"]))
        return
    p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in fill_state(o_ecapa.ecapa_shapes()).items()}
    tr = o_train.OracleTrainer("ecapa", p64, fill_value("center", (1, 256)).double())
    lo, _, _, go, _, _ = tr.loss_and_grads(x.double(), labels)
    np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-4)
    worst = ("", 0.0)
    for k, ref in go.items():
        if ref is None or k in ("attention.2.bias", "attention.3.bias"):
            continue
        r = ref.numpy().ravel()
        err = np.linalg.norm(got[k] - r) / (np.linalg.norm(r) + 1e-30)
        if err > worst[1]:
            worst = (k, err)
        assert err <= 5e-3, (k, err)
    print("fp32 worst relative L2 gradient error", worst)


@pytest.mark.parametrize("dtype", ["bf16", "bf16c", "fp32"])
def test_ecapa_configs2_size_properties(dtype):
    """BASELINE configs[2] at size under pytest: ECAPA-TDNN-512, batch 128, T = 750 (bf16 compute = configs[2]
    itself, fp32 = the reference's arithmetic).  The CPU oracle needs minutes at this size, so the checks are
    the size-independent properties: (a) determinism - two identical steps give bit-identical gradients,
    (b) linearity - backward of 2 x loss doubles every gradient, (c) eval-mode batch independence - scoring
    128 utterances at once equals scoring them 16 at a time, (d) every gradient finite and non-zero."""
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    B, T = 128, 750
    x = synth_feat((B, 60, T), seed=128).cuda()
    labels = (torch.arange(B) % 5 != 0).long().cuda()

    def fresh():
        torch.manual_seed(688)
        m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)  # seeded kaiming init (= reference's)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        return m.cuda().train().set_compute_dtype(dtype), lossm.cuda()

    gs = []
    for scale in (1.0, 1.0, 2.0):
        m, lossm = fresh()
        feat, _ = m(x)
        loss, _ = lossm(feat, labels)
        (scale * loss).backward()
        gs.append((m.arena().grad[:m.arena().head_total].clone(), lossm.center.grad.clone(), loss.item()))
    assert torch.equal(gs[0][0], gs[1][0]) and torch.equal(gs[0][1], gs[1][1]) and gs[0][2] == gs[1][2]
    g1, g2 = gs[0][0], gs[2][0]
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert float((g2 - 2.0 * g1).abs().max()) <= 2e-5 * float(g1.abs().max())
    for k, p in m.named_parameters():
        if k.startswith(("fc7", "bn7")):
            assert p.grad is None  # no gradient under ang_iso (main_train.py:355 -> 376)
        elif k not in ("attention.2.bias", "attention.3.bias"):  # analytically zero (softmax over T)
            assert p.grad is not None and float(p.grad.abs().max()) > 0, k
    m.eval()
    with torch.no_grad():
        f_all, _ = m(x)
        f_parts = torch.cat([m(x[i:i + 16])[0] for i in range(0, B, 16)])
    assert float((f_all - f_parts).abs().max()) <= 2e-5 * float(f_all.abs().max())
