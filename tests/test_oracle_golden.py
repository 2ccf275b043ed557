"""CPU: the oracle restatements (oracle/*.py) against the committed golden
vectors that tests/golden/make_golden.py produced from the real reference."""
import numpy as np
import pytest
import torch

from oracle import ecapa as o_ecapa
from oracle import eer as o_eer
from oracle import lfcc as o_lfcc
from oracle import loss as o_loss
from oracle import pad as o_pad
from oracle import resnet as o_resnet
from oracle import train as o_train
from oracle.filler import fill_state, fill_value, synth_feat, synth_pcm

LFCC_TOL = 2e-5  # abs, on log10/DCT outputs of O(1..30); fp32 FFT + log noise (make_golden prints <= 1e-5)


def test_lfcc_filterbank_and_dct(golden):
    g = golden("lfcc.npz")
    fb = o_lfcc.linear_filterbank()
    assert fb.shape == (257, 20)
    assert int((g["fb"] != 0).sum()) == 486
    np.testing.assert_allclose(fb, g["fb"], atol=2e-6)
    assert (fb[0] == 0).all() and (fb[256] == 0).all()
    np.testing.assert_allclose(o_lfcc.dct2_ortho_matrix(), g["dct"], atol=1e-7)


@pytest.mark.parametrize("ci", range(8))
def test_lfcc_forward_cases(golden, ci):
    g = golden("lfcc.npz")
    B, L = [int(v) for v in g["shape%d" % ci]]
    x = synth_pcm(B, L, seed=ci)
    np.testing.assert_allclose([x.double().sum().item(), x.double().abs().sum().item()], g["xsum%d" % ci], rtol=1e-12)
    xn = x.numpy().copy()
    y = o_lfcc.lfcc_forward(xn, fb=g["fb"], dct=g["dct"])
    assert y.shape == (B, 1 + L // 160, 60)
    np.testing.assert_allclose(y, g["y%d" % ci], atol=LFCC_TOL)
    # input is pre-emphasised in place (feature_extraction.py:106)
    np.testing.assert_allclose(xn[:, :64], g["xmut%d" % ci], atol=1e-7)
    # independent float64 restatement agrees too
    y64 = o_lfcc.lfcc_forward(x.numpy().astype(np.float64), dtype=np.float64, mutate=False)
    np.testing.assert_allclose(y64, g["y%d" % ci], atol=LFCC_TOL)


@pytest.mark.parametrize("name", ["sil", "imp", "sine"])
def test_lfcc_structured(golden, name):
    g = golden("lfcc.npz")
    y = o_lfcc.lfcc_forward(g["x_" + name].copy(), fb=g["fb"], dct=g["dct"])
    np.testing.assert_allclose(y, g["y_" + name], atol=LFCC_TOL)
    if name == "sil":
        assert abs(y[0, 0, 0] - (-30.964)) < 1e-2  # SURVEY §8c silence row


def test_pad_chop(golden):
    g = golden("pad.npz")
    ramp = torch.arange(401 * 60, dtype=torch.float32).reshape(1, 401, 60) / 100.0
    np.testing.assert_array_equal(o_pad.repeat_pad(ramp, 750)[0, :, 0].numpy(), g["rep_rows"])
    np.testing.assert_array_equal(o_pad.zero_pad(ramp, 750)[0, :, 0].numpy(), g["zero_rows"])
    sil = torch.from_numpy(g["silence_row"])
    np.testing.assert_array_equal(o_pad.silence_pad(ramp, 750, sil)[0, :, 1].numpy(), g["sil_rows"])
    long = torch.arange(1000 * 60, dtype=torch.float32).reshape(1, 1000, 60)
    for s, start in enumerate(g["chop_starts"]):
        np.random.seed(s)
        out = o_pad.pad_chop(long, 750)
        assert torch.equal(out, long[:, start:start + 750])
    same = torch.zeros(1, 750, 60)
    assert o_pad.pad_chop(same, 750) is same
    with pytest.raises(ValueError):
        o_pad.pad_chop(ramp, 750, padding="bogus")


@pytest.mark.parametrize("mode", ["mixed", "all0", "all1"])
def test_ocsoftmax(golden, mode):
    g = golden("ocsoftmax.npz")
    feats = torch.from_numpy(g["feats_" + mode]).requires_grad_(True)
    center = torch.from_numpy(g["center"]).requires_grad_(True)
    labels = torch.from_numpy(g["labels_" + mode])
    loss, negs = o_loss.ocsoftmax_forward(feats, center, labels, 0.9, 0.2, 20.0)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss_" + mode], rtol=1e-6)
    np.testing.assert_allclose(negs.detach().numpy(), g["negscores_" + mode], atol=1e-6)
    np.testing.assert_allclose(feats.grad.numpy(), g["gfeat_" + mode], atol=1e-7)
    np.testing.assert_allclose(center.grad.numpy(), g["gcenter_" + mode], atol=1e-6)
    l64, n64, gx, gc = o_loss.ocsoftmax_grads_f64(g["feats_" + mode], g["center"], g["labels_" + mode], 0.9, 0.2, 20.0)
    np.testing.assert_allclose(l64, g["loss_" + mode], rtol=2e-6)
    np.testing.assert_allclose(gx, g["gfeat_" + mode], atol=1e-7)
    np.testing.assert_allclose(gc, g["gcenter_" + mode], atol=1e-6)


def test_eer(golden):
    g = golden("eer.npz")
    e1, t1 = o_eer.compute_eer(g["tgt"], g["non"])
    e2, t2 = o_eer.compute_eer(g["tgt_t"], g["non_t"])
    np.testing.assert_allclose([e1, e2], g["eer"], atol=1e-12)
    np.testing.assert_allclose([t1, t2], g["thr"], atol=1e-12)
    # the six committed score files' EERs, known answers
    np.testing.assert_allclose(100 * g["file_eers"], [0.1968, 0.2276, 0.2366, 4.1476, 4.6610, 4.7172], atol=5e-4)


def _att_T(T):
    for _ in range(3):
        T = (T + 2 - 3) // 2 + 1
    return T


@pytest.mark.parametrize("tag,B,T", [("small", 2, 96), ("full", 2, 750)])
def test_resnet_forward(golden, tag, B, T):
    g = golden("resnet.npz")
    shapes = o_resnet.resnet18_shapes()
    assert len(shapes) == 117
    params = fill_state(shapes)
    assert sum(v.numel() for k, v in params.items() if not o_resnet.is_buffer(k)) == 12450290
    x = synth_feat((B, 1, 60, T), seed=200 + T)
    if tag == "small":
        np.testing.assert_array_equal(x.numpy(), g["x_small"])
    for mode in ("train", "eval"):
        torch.manual_seed(1234)
        noise = 1e-5 * torch.randn(B, _att_T(T), 256)
        upd = {}
        feat, mu = o_resnet.resnet18_forward(params, x, training=(mode == "train"), noise=noise, updates=upd)
        np.testing.assert_allclose(feat.numpy(), g["feat_%s_%s" % (tag, mode)], atol=2e-6)
        np.testing.assert_allclose(mu.numpy(), g["mu_%s_%s" % (tag, mode)], atol=2e-6)
        if mode == "train":
            for k in ("bn1.running_mean", "bn1.running_var", "layer4.1.bn2.running_mean", "bn5.running_var"):
                np.testing.assert_allclose(upd[k].numpy(), g["%s_%s" % (k, tag)], atol=1e-5)


def test_resnet_grads_small(golden):
    g = golden("resnet.npz")
    params = fill_state(o_resnet.resnet18_shapes())
    x = synth_feat((2, 1, 60, 96), seed=296)
    torch.manual_seed(1234)
    noise = 1e-5 * torch.randn(2, 12, 256)
    tr = o_train.OracleTrainer("resnet", params, fill_value("center", (1, 256)))
    taps = {}
    o_resnet.resnet18_forward(params, x, True, noise, None, taps)
    for nm in ("conv1", "layer1", "layer2", "layer3", "layer4", "conv5"):
        np.testing.assert_allclose(taps[nm].double().sum().item(), g["tapsum_" + nm], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(taps[nm].double().abs().sum().item(), g["tapabs_" + nm], rtol=1e-5)
    loss, negs, feat, grads, gcenter, _ = tr.loss_and_grads(x, torch.tensor([0, 1]), noise)
    np.testing.assert_allclose(loss.item(), g["loss_small"], rtol=1e-6)
    assert grads["fc_mu.weight"] is None and grads["fc_mu.bias"] is None
    for k, gr in grads.items():
        if gr is None:
            continue
        np.testing.assert_allclose(gr.norm().item(), g["gnorm_" + k], rtol=2e-4)
    gmax = np.abs(g["g_conv1.weight"]).max()  # filler net is stiff: |g| ~ 1e4, compare relative to max
    np.testing.assert_allclose(grads["conv1.weight"].numpy(), g["g_conv1.weight"], atol=2e-4 * gmax)
    np.testing.assert_allclose(gcenter.numpy(), g["g_center"], rtol=1e-4, atol=1e-6)


def test_adam_matches_torch():
    p = synth_feat((37, 11), 1)
    gr = synth_feat((37, 11), 2)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    mine = p.clone()
    for step in range(1, 4):
        ref.grad = gr * step
        opt.step()
        o_train.adam_step_(mine, gr * step, m, v, step)
        np.testing.assert_allclose(mine.numpy(), ref.detach().numpy(), atol=1e-7)


def test_trajectory(golden):
    g = golden("trajectory.npz")
    params = fill_state(o_resnet.resnet18_shapes())
    tr = o_train.OracleTrainer("resnet", params, fill_value("center", (1, 256)))
    xb = synth_feat((8, 1, 60, 128), seed=300)
    labels = torch.from_numpy(g["labels"])
    losses = []
    for it in range(3):
        torch.manual_seed(500 + it)
        noise = 1e-5 * torch.randn(8, 16, 256)
        losses.append(tr.step(xb, labels, noise)[0].item())
    # step 1 is pre-update (exact); later steps sit on Adam's sign-SGD noise floor (make_golden note)
    np.testing.assert_allclose(losses[0], g["losses"][0], rtol=2e-6)
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-4)
    assert np.abs(tr.params["conv1.weight"].numpy() - g["conv1_w"]).max() <= 3 * 2 * 5e-4 + 1e-6
    np.testing.assert_allclose(tr.center.numpy(), g["center"], atol=1e-6)
    assert int(tr.params["bn1.num_batches_tracked"]) == int(g["nbt"]) == 3


@pytest.mark.parametrize("tag,B,T", [("small", 2, 96), ("full", 2, 750)])
def test_ecapa_forward(golden, tag, B, T):
    g = golden("ecapa.npz")
    shapes = o_ecapa.ecapa_shapes()
    assert len(shapes) == 248
    params = fill_state(shapes)
    assert sum(v.numel() for k, v in params.items() if not o_resnet.is_buffer(k)) == 6337734
    x = synth_feat((B, 60, T), seed=400 + T)
    for mode in ("train", "eval"):
        taps = {}
        feat, out = o_ecapa.ecapa_forward(params, x, training=(mode == "train"), taps=taps)
        np.testing.assert_allclose(feat.numpy(), g["feat_%s_%s" % (tag, mode)], atol=1e-5)
        np.testing.assert_allclose(out.numpy(), g["out_%s_%s" % (tag, mode)], atol=1e-5)
        if mode == "train":
            np.testing.assert_allclose(taps["w"].sum(2).numpy(), 1.0, atol=1e-5)
            np.testing.assert_allclose(taps["mu"].numpy(), g["mu_" + tag], atol=1e-5)
            np.testing.assert_allclose(taps["sg"].numpy(), g["sg_" + tag], atol=1e-5)


def test_ecapa_grads_small(golden):
    g = golden("ecapa.npz")
    params = fill_state(o_ecapa.ecapa_shapes())
    x = synth_feat((2, 60, 96), seed=496)
    tr = o_train.OracleTrainer("ecapa", params, fill_value("center", (1, 256)))
    loss, negs, feat, grads, gcenter, _ = tr.loss_and_grads(x, torch.tensor([0, 1]))
    np.testing.assert_allclose(loss.item(), g["loss_small"], rtol=1e-6)
    for k in ("fc7.weight", "fc7.bias", "bn7.weight", "bn7.bias"):
        assert grads[k] is None
    for k, gr in grads.items():
        if gr is not None:
            np.testing.assert_allclose(gr.norm().item(), g["gnorm_" + k], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(grads["layer2.convs.3.weight"].numpy(), g["g_layer2.convs.3.weight"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("context,summed,enc", [(False, False, "ECA"), (True, True, "ECA"), (False, True, "ECA"),
                                                (True, False, "ASP"), (False, True, "ASP")])
def test_ecapa_constructor_options_pinned_to_the_reference(golden, context, summed, enc):
    """oracle/ecapa.py's ``context=`` / ``summed=`` (ecapa_tdnn.py:126-129, :163-170, :177-180) against the REAL reference
    built with those constructor options (tests/golden/make_golden_ecapa_variants.py -> ecapa_variants.npz)."""
    g = golden("ecapa_variants.npz")
    tag = "c%ss%s" % ("t" if context else "f", "t" if summed else "f") + ("" if enc == "ECA" else "_asp")
    params = fill_state(o_ecapa.ecapa_shapes(context=context, encoder_type=enc))
    x = synth_feat((2, 60, 96), seed=496)
    for mode in ("train", "eval"):
        feat, out = o_ecapa.ecapa_forward(params, x, training=(mode == "train"), context=context, summed=summed)
        np.testing.assert_allclose(feat.numpy(), g["feat_%s_%s" % (tag, mode)], atol=1e-5)
        np.testing.assert_allclose(out.numpy(), g["out_%s_%s" % (tag, mode)], atol=1e-5)
    tr = o_train.OracleTrainer("ecapa", params, fill_value("center", (1, 256)), context=context, summed=summed)
    loss, _, _, grads, gcenter, _ = tr.loss_and_grads(x, torch.tensor([0, 1]))
    np.testing.assert_allclose(loss.item(), g["loss_" + tag], rtol=1e-6)
    for k, gr in grads.items():
        if gr is not None:
            np.testing.assert_allclose(gr.norm().item(), g["gnorm_%s_%s" % (tag, k)], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(grads["conv1.bias"].numpy(), g["g_%s_conv1.bias" % tag], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(grads["layer2.conv1.weight"][:8].numpy(), g["g_%s_layer2.conv1.weight" % tag], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(grads["attention.0.weight"][:4].numpy(), g["g_%s_attention.0.weight" % tag], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(gcenter.numpy(), g["g_%s_center" % tag], rtol=1e-3, atol=1e-6)


def test_ecapa_bf16_oracle_pinned_to_fp32_goldens(golden):
    """BASELINE configs[2] (bf16 compute) has no reference implementation: the bf16 oracle is
    pinned through the reference's fp32 goldens at bf16 tolerance (eval mode: relative L2 of the
    embedding 2.6e-3 measured, bound 1e-2), and its hand-written backward is checked against
    autograd of the same rounded-operand contraction."""
    g = golden("ecapa.npz")
    params = fill_state(o_ecapa.ecapa_shapes())
    x = synth_feat((2, 60, 96), seed=496)
    feat, out = o_ecapa.ecapa_forward(params, x, training=False, bf16=True)
    ref = g["feat_small_eval"]
    rel = np.linalg.norm(feat.numpy() - ref) / np.linalg.norm(ref)
    assert 1e-4 < rel < 1e-2, rel
    xx = synth_feat((2, 32, 20), 1).double()
    ww = synth_feat((16, 32, 1), 2).double()
    dy = synth_feat((2, 16, 20), 3).double()
    xa, wa = xx.clone().requires_grad_(True), ww.clone().requires_grad_(True)
    o_ecapa._Bf16Pointwise.apply(xa, wa).backward(dy)
    rnd = lambda t: t.to(torch.bfloat16).double()
    xb, wb = rnd(xx).requires_grad_(True), rnd(ww).requires_grad_(True)
    torch.nn.functional.conv1d(xb, wb).backward(rnd(dy))
    np.testing.assert_allclose(xa.grad.numpy(), xb.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(wa.grad.numpy(), wb.grad.numpy(), atol=1e-12)


def _bf16_ulp(a):
    """Spacing of bf16 values at |a| (8 significant bits)."""
    return np.exp2(np.floor(np.log2(np.maximum(np.abs(a), 1e-30))) - 7)


@pytest.mark.parametrize("mode", [True, "resident"])
def test_ecapa_bf16_oracle_vs_reference_under_autocast(golden, mode):
    """X2 pin: tests/golden/ecapa_bf16.npz holds the REAL reference ``Res2Net2`` run under
    ``torch.autocast('cpu', bfloat16)`` on the filler weights (make_golden_bf16.py).

    TRAIN mode (round 3): on this filler-initialised net batch statistics amplify bf16 rounding ~100x, so two bf16
    arithmetics agree only statistically - autocast itself sits 0.14 (B = 2, T = 96) relative L2 from the fp32
    reference on the embedding.  Asserted, all computed HERE from the oracle's output and the reference's tensors:
    the oracle is no farther from the autocast run than the autocast run is from fp32 (x 1.25), its loss inside the
    same band.

    EVAL mode (round 4, running statistics: no amplification) - the sharp part:
    * the bf16 NOISE FLOOR: the reference under autocast sits 1.08e-2 (2, 96) / 1.19e-2 (8, 750) from its own fp32
      run on the embedding, 6.5e-3 already behind the first Bottle2neck - two independent bf16 evaluations of this
      graph cannot be closer than that to each other; the oracle must lie inside 1.25e-2 of the autocast run
      (measured 1.03e-2 / 1.12e-2) AND closer to the fp32 reference than autocast is (<= 6e-3; measured 4.6e-3 /
      4.2e-3: statistics and biases are kept in fp32 here);
    * where the rule sets COINCIDE the pin is bit-level: the first layer's stored output (conv1 -> ReLU -> bn1,
      ecapa_tdnn.py:159-161) of the resident mode reproduces the autocast run's bf16 BITS on all 49,152 values of
      utterance 0 once the conv bias is rounded the way autocast rounds it (``oracle.ecapa.AUTOCAST_BIAS``), and with
      this build's fp32 bias at most 5 % of the values move (3.1 %; relative L2 of the tensor 8.9e-4)."""
    g = golden("ecapa_bf16.npz")
    seed, B, T = [int(v) for v in g["x_seed_small"]]
    x = synth_feat((B, 60, T), seed=seed)
    labels = (torch.arange(B) % 3 != 0).long()
    tr = o_train.OracleTrainer("ecapa", fill_state(o_ecapa.ecapa_shapes()), fill_value("center", (1, 256)), bf16=mode)
    loss, _, feat, grads, _, _ = tr.loss_and_grads(x, labels)
    fa, f32 = g["feat_autocast_small"].astype(np.float64), g["feat_fp32_small"].astype(np.float64)
    d_auto = np.linalg.norm(fa - f32) / np.linalg.norm(f32)
    d_mine = np.linalg.norm(feat.double().numpy() - fa) / np.linalg.norm(fa)
    assert d_mine <= 1.25 * d_auto, (d_mine, d_auto)
    la, l32 = float(g["loss_autocast_small"]), float(g["loss_fp32_small"])
    assert abs(loss.item() - la) <= 3.0 * abs(la - l32) + 1e-3 * l32, (loss.item(), la, l32)
    # ---- eval mode
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    params = fill_state(o_ecapa.ecapa_shapes())
    for tag in ("small", "full"):
        seed, B, T = [int(v) for v in g["x_seed_" + tag]]
        xe = synth_feat((B, 60, T), seed=seed)
        taps = {}
        fe, _ = o_ecapa.ecapa_forward(params, xe, training=False, bf16=mode, taps=taps)
        fae, f32e = g["feat_autocast_eval_" + tag].astype(np.float64), g["feat_fp32_eval_" + tag].astype(np.float64)
        floor = rel(fae, f32e)
        assert 5e-3 < floor < 1.5e-2            # the reference's own bf16 noise floor (1.08e-2 / 1.19e-2)
        assert rel(fe.numpy(), fae) <= 1.25e-2, (tag, rel(fe.numpy(), fae))
        assert rel(fe.numpy(), f32e) <= 6e-3, (tag, rel(fe.numpy(), f32e))
        if tag == "small" and mode == "resident":
            x1a, x1f = g["x1_sample_autocast_eval_small"], g["x1_sample_fp32_eval_small"].astype(np.float64)
            assert rel(taps["x1"][0, ::4].numpy(), x1a.astype(np.float64)) <= 7e-3     # measured 5.6e-3 (floor 6.5e-3)
            assert rel(taps["x1"][0, ::4].numpy(), x1f) <= 7e-3
            want = torch.from_numpy(g["h0_bits_autocast_eval_small"].view(np.int16)).view(torch.bfloat16).float().numpy()
            got = taps["h0"][0].numpy()
            diff = got != want
            assert diff.mean() <= 0.05, diff.mean()                                     # measured 3.1 %
            # each moved value by one ulp of the CONV output it came from (the BatchNorm behind it scales that and can
            # cancel against beta, so "ulps of the stored value" is not the measure): relative L2 8.9e-4, no entry by
            # more than one ulp of the tensor's largest value
            assert rel(got, want.astype(np.float64)) <= 1.5e-3
            assert np.abs(got - want).max() <= _bf16_ulp(np.abs(want).max())
            o_ecapa.AUTOCAST_BIAS = True
            try:
                t2 = {}
                o_ecapa.ecapa_forward(params, xe, training=False, bf16=mode, taps=t2)
            finally:
                o_ecapa.AUTOCAST_BIAS = False
            assert int((t2["h0"][0].numpy() != want).sum()) <= 8                        # measured 0 of 49,152
