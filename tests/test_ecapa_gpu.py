"""GPU parity of the drop-in ECAPA-TDNN (Res2Net2) against the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import ecapa as o_ecapa
from oracle import train as o_train
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat

from _budget import check_bf16_band, record  # noqa: E402

pytestmark = pytest.mark.gpu


def make_model():
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    fill_module_(m)
    return m.cuda()


def test_state_dict_surface():
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    want = o_ecapa.ecapa_shapes()
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got.items()) == [(k, tuple(v)) for k, v in want.items()]
    assert sum(p.numel() for p in m.parameters()) == 6337734


@pytest.mark.parametrize("tag,B,T", [("small", 2, 96), ("full", 2, 750)])
def test_forward_vs_golden(golden, tag, B, T):
    g = golden("ecapa.npz")
    m = make_model()
    x = synth_feat((B, 60, T), seed=400 + T)
    for mode in ("train", "eval"):
        fill_module_(m)
        m.train(mode == "train")
        with torch.no_grad():
            feat, out = m(x.cuda())
        # |feat| ~ 3; fp32 summation-order noise through ~40 layers
        np.testing.assert_allclose(feat.cpu().numpy(), g["feat_%s_%s" % (tag, mode)], atol=2e-4)
        np.testing.assert_allclose(out.cpu().numpy(), g["out_%s_%s" % (tag, mode)], atol=5e-4)


@pytest.mark.parametrize("B,T,tol", [(2, 96, 5e-2), (8, 64, 5e-3)])
def test_grads_vs_oracle_small(golden, B, T, tol):
    """All 150 gradients vs the fp64 oracle, as relative L2 error per tensor.
    Two effects bound what any fp32 implementation can match (both measured on the CPU oracle
    itself, fp32 vs fp64): (1) B = 2 (the golden case, loss pinned to the reference) has
    2-sample BatchNorms in the SE blocks and is stiff: fp32-CPU moves gradients by 3e-2 of max;
    (2) a pre-activation within rounding of 0 can land on the other side of a ReLU: one such
    flip among 786k layer4 outputs changes that channel's bias gradient by a whole element
    (seen: 1 flip -> 1.3 % of max on layer4.bias while every elementwise gradient agreed to
    4e-5).  Hence L2 per tensor, with a loose max bound."""
    g = golden("ecapa.npz")
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    m = make_model().train()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    x = synth_feat((B, 60, T), seed=400 + T)
    labels = (torch.arange(B) % 3 != 0).long()
    feat, out = m(x.cuda())
    loss, _ = lossm(feat, labels.cuda())
    loss.backward()
    if (B, T) == (2, 96):
        np.testing.assert_allclose(loss.item(), g["loss_small"], rtol=1e-4)
    # fp64 oracle: with B = 2 the batch statistics are stiff and an fp32 CPU evaluation of the
    # same graph already moves first-layer gradients by ~3e-3 of max
    p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in fill_state(o_ecapa.ecapa_shapes()).items()}
    tr = o_train.OracleTrainer("ecapa", p64, fill_value("center", (1, 256)).double())
    lo, no, fo, go, gco, _ = tr.loss_and_grads(x.double(), labels)
    worst = ("", 0.0)
    # attention.2.bias and attention.3.bias have analytically ZERO gradients (they shift every
    # logit of a (b, c) row by the same amount and softmax over T ignores that), so both sides
    # hold rounding noise there: bound it against the gradient scale of the sibling gamma.
    noise_floor = 1e-4 * float(go["attention.2.weight"].abs().max())
    for k, p in m.named_parameters():
        if go[k] is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        ref = go[k].numpy()
        got = p.grad.cpu().double().numpy()
        diff = np.abs(got - ref).max()
        err = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        if k in ("attention.2.bias", "attention.3.bias"):
            assert diff < noise_floor, (k, diff, noise_floor)
            continue
        assert diff <= 10 * tol * np.abs(ref).max(), (k, diff, np.abs(ref).max())
        if err > worst[1]:
            worst = (k, err)
        assert err < tol, "%s: relative L2 grad err %.3g (|ref|max %.3g)" % (k, err, np.abs(ref).max())
    print("worst relative L2 grad err", worst)


def test_ce_branch_gradients():
    """With a loss on ``out`` too (base-loss branch, main_train.py:355) fc7/bn7 receive gradients."""
    m = make_model().train()
    x = synth_feat((4, 60, 64), seed=3)
    wo, wf = synth_feat((4, 2), 9), synth_feat((4, 256), 10)  # non-degenerate readout
    feat, out = m(x.cuda())
    ((feat * wf.cuda()).sum() * 0.01 + (out * wo.cuda()).sum()).backward()
    assert m.fc7.weight.grad is not None and m.bn7.weight.grad is not None
    tr = o_train.OracleTrainer("ecapa", fill_state(o_ecapa.ecapa_shapes()), fill_value("center", (1, 256)))
    names = tr.trainable()
    for k in names:
        tr.params[k] = tr.params[k].detach().requires_grad_(True)
    fo, oo = tr.forward(x)
    ((fo * wf).sum() * 0.01 + (oo * wo).sum()).backward()
    for k in ("fc7.weight", "bn7.bias", "layer2.convs.3.weight", "conv1.weight", "attention.0.weight"):
        ref = tr.params[k].grad.numpy()
        got = dict(m.named_parameters())[k].grad.cpu().numpy()
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < 2e-3, (k, err)


# ------------------------------------------------------------------ bf16 compute (BASELINE configs[2])
BF16_MODES = [("bf16c", True), ("bf16", "resident")]  # (HIP compute_dtype, oracle bf16 mode)


@pytest.mark.parametrize("hip_dt,omode", BF16_MODES)
def test_bf16_forward_vs_bf16_oracle_and_fp32_golden(golden, hip_dt, omode):
    """Eval-mode forward in bf16 compute vs (a) the bf16 oracle (same arithmetic: bf16-rounded
    operands, fp32 accumulation) and (b) the reference's fp32 goldens at bf16 tolerance.
    A value within fp32 noise of a bf16 rounding boundary can round the other way on the GPU
    (different summation order); each such flip moves one operand by 2^-9 relative and the Res2 chains
    stack 21 bf16 convs in sequence, so (a) is bounded at 3e-3 of |feat|max (measured 1.4e-3) instead of
    the fp32 path's 1e-4.  The arithmetic itself is pinned at 2e-5 per kernel (test_conv1d_bf16_gpu.py)."""
    g = golden("ecapa.npz")
    m = make_model().eval().set_compute_dtype(hip_dt)
    params = fill_state(o_ecapa.ecapa_shapes())
    # resident: every activation is ALSO rounded when stored (a flip there moves a value by 2^-8 relative), and
    # twice as many tensors are rounded: 6e-3 of |feat|max; the embedding stays within 2e-2 of the fp32 golden
    tol_o, tol_g = (3e-3, 1e-2) if omode is True else (6e-3, 2e-2)
    for tag, B, T in (("small", 2, 96), ("full", 2, 750)):
        x = synth_feat((B, 60, T), seed=400 + T)
        with torch.no_grad():
            feat, out = m(x.cuda())
        fo, oo = o_ecapa.ecapa_forward(params, x, training=False, bf16=omode)
        scale = float(fo.abs().max())
        print(hip_dt, tag, "feat vs oracle", float((feat.cpu() - fo).abs().max()) / scale)
        assert float((feat.cpu() - fo).abs().max()) <= tol_o * scale
        assert float((out.cpu() - oo).abs().max()) <= tol_o * max(float(oo.abs().max()), 1.0)
        ref = g["feat_%s_eval" % tag]
        rel = np.linalg.norm(feat.cpu().numpy() - ref) / np.linalg.norm(ref)
        assert rel <= tol_g, rel  # measured 2.6e-3 (compute only): the 2^-9 operand rounding through ~20 layers
        # and the fp32 mode of the same module is still the reference's arithmetic
    m.set_compute_dtype("fp32")
    with torch.no_grad():
        feat, _ = m(synth_feat((2, 60, 96), seed=496).cuda())
    np.testing.assert_allclose(feat.cpu().numpy(), g["feat_small_eval"], atol=2e-4)


def test_bf16_eval_forward_vs_reference_under_autocast(golden):
    """X2 pin on the GPU: the HIP eval forward in bf16-resident mode against the REAL reference
    ``Res2Net2`` run under ``torch.autocast('cpu', bfloat16)`` in eval mode (tests/golden/make_golden_bf16.py,
    ``*_autocast_eval_*``) - not against the fp32 golden.  Running statistics: no amplification.  The reference's own
    bf16 noise floor (autocast vs its fp32 run) is 1.08e-2 at (2, 96) and 1.19e-2 at (8, 750) on the embedding, so
    no independent bf16 evaluation can be asked to sit closer to the autocast run than that: <= 1.25e-2 of it
    (oracle: 1.03e-2 / 1.12e-2), and <= 6e-3 of the fp32 reference (statistics and biases stay fp32 here: closer to
    fp32 than autocast is).  Bit-level where the rule sets coincide - the first layer's stored output: identical to
    the resident oracle's on >= 99.9 % of the values (fp32 summation order on rounding boundaries), hence at most 5 %
    of the values away from the autocast run's bits (the fp32 bias, oracle/ecapa.py::AUTOCAST_BIAS)."""
    from asvspoof2021_air_amd import ops_h as oh
    g = golden("ecapa_bf16.npz")
    m = make_model().eval().set_compute_dtype("bf16")
    params = fill_state(o_ecapa.ecapa_shapes())
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    for tag in ("small", "full"):
        seed, B, T = [int(v) for v in g["x_seed_" + tag]]
        x = synth_feat((B, 60, T), seed=seed)
        with torch.no_grad():
            feat, _ = m(x.cuda())
        fae, f32e = g["feat_autocast_eval_" + tag].astype(np.float64), g["feat_fp32_eval_" + tag].astype(np.float64)
        d_a, d_f = rel(feat.cpu().numpy(), fae), rel(feat.cpu().numpy(), f32e)
        record("ecapa_bf16_eval_vs_autocast[%s]" % tag, {"vs_autocast": d_a, "vs_fp32_reference": d_f,
                                                          "autocast_vs_fp32": rel(fae, f32e)})
        assert d_a <= 1.25e-2, (tag, d_a)
        assert d_f <= 6e-3, (tag, d_f)
    # first layer, utterance 0 of the (2, 96) case, through the model's own resident-path calls (ecapa_tdnn.py::_forward_h)
    seed, B, T = [int(v) for v in g["x_seed_small"]]
    x = synth_feat((B, 60, T), seed=seed)
    with torch.no_grad():
        xcol = oh.unfold(x.cuda().contiguous(), m.conv1.kernel_size[0], 1, m.conv1.padding[0], m._conv1_rows())
        r0 = oh.conv_pointwise(xcol, m._conv1_matrix(), T, bias=m.conv1.bias.detach(), relu=True)
        st0 = m._bn_h(r0, T, m.bn1, False)
        h = oh.bn_apply(r0, T, st0[2], st0[3])
    got = h[0, :, :T].contiguous().view(torch.bfloat16).float().cpu().numpy()
    taps = {}
    o_ecapa.ecapa_forward(params, x, training=False, bf16="resident", taps=taps)
    same = float((got == taps["h0"][0].numpy()).mean())
    want = torch.from_numpy(g["h0_bits_autocast_eval_small"].view(np.int16)).view(torch.bfloat16).float().numpy()
    moved = float((got != want).mean())
    record("ecapa_bf16_first_layer_bits", {"identical_to_oracle": same, "differ_from_autocast": moved})
    assert same >= 0.999, same
    assert moved <= 0.05 and rel(got, want.astype(np.float64)) <= 1.5e-3, (moved, rel(got, want.astype(np.float64)))


@pytest.mark.parametrize("hip_dt,omode", BF16_MODES)
def test_bf16_grads_vs_bf16_oracle(hip_dt, omode):
    """All gradients of one bf16-compute train step vs the fp64 evaluation of the bf16 oracle.
    Tolerance: this filler-initialised net amplifies perturbations ~100x and bf16 rounding is
    discontinuous (a value within fp32 noise of a rounding boundary rounds the other way), so the ORACLE
    ITSELF - fp32 vs fp64 evaluation of the same bf16 graph - moves gradients by 0.13 median / 0.22 max
    relative L2 per tensor at this size (fp32 graph: 6e-4 / 3e-3).  The band is measured in the test
    (see the assertions); loss rtol 2e-3.  The tight check of the bf16 arithmetic is tests/test_conv1d_bf16_gpu.py (2e-5)."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    B, T = 32, 96
    m = make_model().train().set_compute_dtype(hip_dt)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    x = synth_feat((B, 60, T), seed=400 + T)
    labels = (torch.arange(B) % 3 != 0).long()
    feat, _ = m(x.cuda())
    loss, _ = lossm(feat, labels.cuda())
    loss.backward()
    got = {k: p.grad.cpu().double().numpy().ravel() for k, p in m.named_parameters() if p.grad is not None}
    band, errs = o_train.bf16_gradient_band(x, labels, got, omode)
    lo = band["loss64"]
    np.testing.assert_allclose(loss.item(), lo, rtol=2e-3)
    check_bf16_band(errs, band)
    print("bf16 grads vs bf16 oracle (fp64): median rel L2 %.3g, max %.3g; oracle fp32-vs-fp64 band: median %.3g max %.3g"
          % (np.median([e for e, _ in errs.values()]), max(e for e, _ in errs.values()), band["median"], band["max"]))


def test_bf16_training_tracks_fp32():
    """Three optimisation steps in bf16 compute follow the fp32 run of the same module: the loss
    falls and stays within 5 % of the fp32 trajectory (BASELINE north_star: training loss within
    stated tolerance)."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    x = synth_feat((16, 60, 128), seed=77).cuda()
    labels = (torch.arange(16) % 3 != 0).long().cuda()
    curves = {}
    for dt in ("fp32", "bf16", "bf16c"):
        m = make_model().set_compute_dtype(dt)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        tr = Trainer(m, loss_module=lossm, feat_len=128, ecapa=True)
        curves[dt] = [tr.step_features(x, labels)[0].item() for _ in range(3)]
    print(curves)
    assert curves["bf16"][2] < curves["bf16"][0]
    np.testing.assert_allclose(curves["bf16"], curves["fp32"], rtol=5e-2)
    np.testing.assert_allclose(curves["bf16c"], curves["fp32"], rtol=5e-2)


def test_whole_module_pickle_roundtrip(tmp_path):
    m = make_model().train().set_compute_dtype("bf16")
    x = synth_feat((4, 60, 64), seed=5).cuda()
    feat, _ = m(x)
    feat.sum().backward()
    torch.save(m, tmp_path / "ecapa.pt")
    m2 = torch.load(tmp_path / "ecapa.pt", weights_only=False)
    assert m2.compute_dtype == "bf16"
    m.eval()
    m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x)[0], m2(x)[0])


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16c"])
def test_gradient_accumulation_two_backwards(dtype):
    """backward twice without zero_grad: p.grad (a view of the gradient arena) must end as the SUM of both
    gradients, like torch's AccumulateGrad - not twice the second one (ECAPA had no guard)."""
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    torch.manual_seed(688)
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60).cuda().train().set_compute_dtype(dtype)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0).cuda()
    xa, xb = synth_feat((4, 60, 96), seed=1).cuda(), synth_feat((4, 60, 96), seed=2).cuda()
    labels = torch.tensor([0, 1, 1, 0]).cuda()

    def grads_of(x):
        for p in m.parameters():
            p.grad = None
        m.bn1.running_mean.zero_()  # (forward state does not matter for the gradients; keep runs comparable)
        feat, _ = m(x)
        lossm(feat, labels)[0].backward()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    ga, gb = grads_of(xa), grads_of(xb)
    for p in m.parameters():
        p.grad = None
    for x in (xa, xb):
        feat, _ = m(x)
        lossm(feat, labels)[0].backward()
    arena = m.arena()
    for k, p in m.named_parameters():
        if k not in ga:
            continue
        want = ga[k] + gb[k]
        assert p.grad.data_ptr() == arena.grad_view(k).data_ptr()  # still the zero-copy arena view
        tol = 1e-6 * float(want.abs().max()) + 1e-12
        assert float((p.grad - want).abs().max()) <= tol, k


@pytest.mark.parametrize("training", [True, False])
def test_bottle2neck_and_se_standalone_forward(training):
    """Bottle2neck.forward / SEModule.forward on their own (ecapa_tdnn.py:64-95, :27-29) against the oracle."""
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, SEModule
    blk = Bottle2neck(512, 512, kernel_size=3, dilation=3, scale=8)
    fill_module_(blk)
    sd = {"b." + k: v.clone() for k, v in blk.state_dict().items()}
    blk = blk.cuda().train(training)
    x = synth_feat((4, 512, 96), seed=62)
    with torch.no_grad():
        got = blk(x.cuda())
    want = o_ecapa.bottle2neck(x, sd, "b", 3, 8, training, {})
    assert float((got.cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max())
    se = SEModule(512)
    fill_module_(se)
    sds = {"s." + k: v.clone() for k, v in se.state_dict().items()}
    se = se.cuda().train(training)
    with torch.no_grad():
        got = se(x.cuda())
    want = o_ecapa.se_module(x, sds, "s", training, {})
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    if training:
        with pytest.raises(NotImplementedError):
            blk(x.cuda())


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_graphed_train_step_equals_eager(dtype):
    """Trainer.enable_graph(): the hipGraph replay of front-end + forward + loss + backward (optimisers outside) ends
    on bit-identical weights, centre and BatchNorm statistics as the eager launches, over five steps with changing
    batches and a learning-rate change in between."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    from oracle.filler import synth_pcm
    batches = [(synth_pcm(8, 16000, seed=300 + i).cuda(), ((torch.arange(8) + i) % 3 != 0).long().cuda()) for i in range(5)]
    ends = []
    for graph in (False, True):
        m = make_model().set_compute_dtype(dtype)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        tr = Trainer(m, loss_module=lossm, feat_len=128, ecapa=True)
        if graph:
            tr.enable_graph()
        losses = []
        for i, (pcm, lab) in enumerate(batches):
            if i == 3:
                tr.set_epoch(4, lr_decay=0.5, interval=4)
            losses.append(tr.step(pcm, lab)[0].item())
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        ends.append((losses, m.arena().flat.clone(), tr.loss.center.detach().clone(), m.bn1.running_var.clone(),
                     int(m.bn1.num_batches_tracked)))
    (l0, w0, c0, rv0, n0), (l1, w1, c1, rv1, n1) = ends
    assert l0 == l1 and torch.equal(w0, w1) and torch.equal(c0, c1) and torch.equal(rv0, rv1) and n0 == n1 == 5


def test_graphed_steps_interleaved_with_eager():
    """an eager step, an external zero_grad() or a score() call between replays must not detach
    the optimisers from the gradients the captured kernels write (the loss centre's p.grad lives in the graph's private
    pool; zero_grad() drops every p.grad).  Same sequence eager-only and with the graph enabled -> bit-identical ends;
    the returned -scores of a replay survive the next replay (they used to alias the graph's static output)."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    from oracle.filler import synth_pcm
    batches = [(synth_pcm(8, 16000, seed=400 + i).cuda(), ((torch.arange(8) + i) % 3 != 0).long().cuda()) for i in range(11)]
    big = (synth_pcm(24, 16000, seed=77).cuda(), (torch.arange(24) % 4 != 0).long().cuda())
    ends = []
    for graph in (False, True):
        m = make_model().set_compute_dtype("bf16")
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        tr = Trainer(m, loss_module=lossm, feat_len=128, ecapa=True)
        if graph:
            tr.enable_graph()
        losses, kept = [], None
        for i, (pcm, lab) in enumerate(batches):
            if i == 3:    # eager step on the same Trainer (what bench.py's roofline leg and step(start=...) do)
                out = tr.step_features(tr.features(pcm), lab)
            elif i == 5:  # external zero_grad + a scoring pass in between
                tr.feat_optimizer.zero_grad(); tr.loss_optimizer.zero_grad()
                tr.score(pcm)
                out = tr.step(pcm, lab)
            elif i == 6:  # a larger eager batch outgrows the scratch buffers the graph points into
                out = tr.step_features(tr.features(big[0]), big[1])
            else:
                out = tr.step(pcm, lab)
            if i == 4:
                kept = (out[1], out[1].clone())
            losses.append(out[0].item())
        torch.cuda.synchronize()
        assert torch.equal(kept[0], kept[1]), "a replay overwrote the scores returned by an earlier step"
        ends.append((losses, m.arena().flat.clone(), tr.loss.center.detach().clone(), m.bn1.running_var.clone()))
        if graph:
            assert tr._graph is not None and tr.model.training
    (l0, w0, c0, rv0), (l1, w1, c1, rv1) = ends
    assert l0 == l1 and torch.equal(w0, w1) and torch.equal(c0, c1) and torch.equal(rv0, rv1)


def test_other_widths():
    """Widths other than the reference's C = 512: C = 1024 trains in both bf16 modes (the legacy mode's
    two-operand dgrad epilogue is chosen per capability, not assumed) and in fp32; C = 256 (32-channel Res2 branches:
    below the 64-channel tiles of every conv1d kernel of this build) is refused loudly, not silently mis-computed."""
    from asvspoof2021_air_amd import _hip
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    x = synth_feat((4, 60, 64), seed=9).cuda()
    for C, dts in ((1024, ("bf16", "bf16c", "fp32")),):
        for dt in dts:
            torch.manual_seed(688)
            m = Res2Net2(Bottle2neck, C=C, model_scale=8, nOut=2, n_mels=60).cuda().train().set_compute_dtype(dt)
            feat, _ = m(x)
            feat.square().mean().backward()
            g = m.layer2.conv1.weight.grad
            assert torch.isfinite(feat).all() and torch.isfinite(g).all() and float(g.abs().max()) > 0, (C, dt)
    m = Res2Net2(Bottle2neck, C=256, model_scale=8, nOut=2, n_mels=60).cuda().train().set_compute_dtype("bf16")
    with pytest.raises(_hip.AirError, match="bf16c"):
        m(x)
    with pytest.raises(_hip.AirError):
        m.set_compute_dtype("fp32")(x)


def test_bf16_resident_long_input_falls_back_to_bf16c():
    """beyond the bf16-resident kernels' row length (oh.max_tp() frames) a 'bf16' model used to
    RAISE and the caller had to switch dtype by hand; now that call runs as 'bf16c' (bf16 matrix cores on fp32 tensors,
    any length like the reference) behind a one-time warning, forward and backward, bit-identical to a model that was
    set to 'bf16c' explicitly - and a short input afterwards takes the resident path again."""
    import warnings
    from asvspoof2021_air_amd import ops_h as oh
    T = oh.max_tp() + 40
    x = synth_feat((2, 60, T), seed=31).cuda()
    outs = {}
    for dt in ("bf16", "bf16c"):
        m = make_model().train().set_compute_dtype(dt)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            feat, _ = m(x)
            feat.square().mean().backward()
            feat2, _ = m(x)  # the warning is issued once
        assert (len([r for r in w if "bf16c" in str(r.message)]) == (1 if dt == "bf16" else 0)), [str(r.message) for r in w]
        outs[dt] = (feat.detach().clone(), m.layer2.conv1.weight.grad.detach().clone())
        if dt == "bf16":
            assert m.compute_dtype == "bf16"
            short = synth_feat((2, 60, 96), seed=32).cuda()
            f_short, _ = m(short)
            m2 = make_model().train().set_compute_dtype("bf16")
            f_want, _ = m2(short)
            # (the first model's BatchNorm buffers moved on; train-mode outputs do not read them)
            assert torch.equal(f_short, f_want)
    assert torch.equal(outs["bf16"][0], outs["bf16c"][0]) and torch.equal(outs["bf16"][1], outs["bf16c"][1])


@pytest.mark.parametrize("context,summed,enc", [(False, False, "ECA"), (True, True, "ECA"), (False, True, "ECA"),
                                                (True, False, "ASP"), (False, True, "ASP")])
def test_non_default_constructor_options_vs_the_reference(golden, context, summed, enc):
    """VERDICT r5 missing 3: ``Res2Net2(context=False)`` (ecapa_tdnn.py:126-129, :177-180) and ``summed=True`` (:163-166)
    - the variants the reference's own score files were made with (lfcc_ecapa512c{t,f}s{t,f}_*) - and ``encoder_type='ASP'``
    (:133-134: one attention weight per frame; served by the ECA kernels on the repeated weight row) used to raise.  fp32
    path against tests/golden/ecapa_variants.npz (the REAL reference, make_golden_ecapa_variants.py): the state_dict
    surface, train / eval forward, the OC-Softmax loss and every gradient - norms against the reference's, tensors
    against the fp64 oracle at the bounds of test_grads_vs_oracle_small (B = 2 is stiff: 5e-2; B = 8: 5e-3); the bf16
    modes refuse these options by name."""
    from asvspoof2021_air_amd import _hip
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    g = golden("ecapa_variants.npz")
    tag = "c%ss%s" % ("t" if context else "f", "t" if summed else "f") + ("" if enc == "ECA" else "_asp")
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60, context=context, summed=summed, encoder_type=enc)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [
        (k, tuple(v)) for k, v in o_ecapa.ecapa_shapes(context=context, encoder_type=enc).items()]
    fill_module_(m)
    m = m.cuda()
    x = synth_feat((2, 60, 96), seed=400 + 96)
    sh = o_ecapa.ecapa_shapes(context=context, encoder_type=enc)
    for mode in ("train", "eval"):
        fill_module_(m)
        m.train(mode == "train")
        with torch.no_grad():
            feat, out = m(x.cuda())
        # The golden is the fp32 reference.  Under 'ASP' with the filler weights the one attention row is nearly one-hot
        # over time (largest weight 0.99999), so sg = sqrt(clamp(sum x^2 w - mu^2, 1e-4)) (:185) is a difference of two
        # numbers of ~48 that agree to 1e-4: any two fp32 summation orders differ by 1e-4 .. 1e-3 there (the HIP mu sits
        # 4e-5 from torch's on values of 7).  The bound is therefore the default test's constant or 4 x the distance of
        # the ORACLE's own fp32 evaluation from its fp64 one, whichever is larger.
        p32 = fill_state(sh)
        p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in p32.items()}
        f32, o32 = o_ecapa.ecapa_forward(p32, x, training=(mode == "train"), context=context, summed=summed)
        f64, o64 = o_ecapa.ecapa_forward(p64, x.double(), training=(mode == "train"), context=context, summed=summed)
        bf, bo = float((f32.double() - f64).abs().max()), float((o32.double() - o64).abs().max())
        np.testing.assert_allclose(feat.cpu().numpy(), g["feat_%s_%s" % (tag, mode)], atol=max(2e-4, 4 * bf))
        np.testing.assert_allclose(out.cpu().numpy(), g["out_%s_%s" % (tag, mode)], atol=max(5e-4, 4 * bo))
        np.testing.assert_allclose(feat.cpu().double().numpy(), f64.numpy(), atol=max(2e-4, 4 * bf))
    # 'ASP' gradients: with the filler weights the single attention row is one-hot over time and sg sits on its clamp -
    # the oracle's own fp32 evaluation is 9 % from its fp64 one there, which pins nothing.  The gradient part therefore
    # runs on a TEMPERED attention (attention.3.weight x 0.02 in the model and in the oracle: weights spread over the
    # frames); the golden comparison of loss / norms applies to the untempered variants only.
    temper = 0.02 if enc == "ASP" else 1.0
    for B, T, tol in ((2, 96, 5e-2), (8, 64, 5e-3)):
        fill_module_(m)
        with torch.no_grad():
            m.attention[3].weight.mul_(temper)
        m.train()
        m.zero_grad(set_to_none=True)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        lossm = lossm.cuda()
        xx = synth_feat((B, 60, T), seed=400 + T)
        labels = torch.tensor([0, 1]) if B == 2 else (torch.arange(B) % 3 != 0).long()
        feat, _ = m(xx.cuda())
        loss, _ = lossm(feat, labels.cuda())
        loss.backward()
        p64 = {k: (v.double() if v.dtype.is_floating_point else v)
               for k, v in fill_state(o_ecapa.ecapa_shapes(context=context, encoder_type=enc)).items()}
        p64["attention.3.weight"] = p64["attention.3.weight"] * temper
        tr = o_train.OracleTrainer("ecapa", p64, fill_value("center", (1, 256)).double(), context=context, summed=summed)
        lo, _, _, go, gco, _ = tr.loss_and_grads(xx.double(), labels)
        # The summed variants are stiffer than the default graph: the ORACLE ITSELF, fp32 against fp64, moves gradients
        # by 3.7e-4 (context, summed) / 5.3e-3 (no context, summed) relative L2 at B = 8 where the default graph moves
        # 5e-6 (the block inputs x + x1 + x2 grow, more pre-activations sit within rounding of a ReLU): the bound on the
        # median tensor scales with the oracle's own fp32 distance, measured here.
        p32 = fill_state(o_ecapa.ecapa_shapes(context=context, encoder_type=enc))
        p32["attention.3.weight"] = p32["attention.3.weight"] * temper
        g32 = o_train.OracleTrainer("ecapa", p32, fill_value("center", (1, 256)), context=context,
                                    summed=summed).loss_and_grads(xx, labels)[3]
        band = {k: float(np.linalg.norm(g32[k].double().numpy() - go[k].numpy()) / (np.linalg.norm(go[k].numpy()) + 1e-30))
                for k in go if go[k] is not None}
        np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-4)
        if B == 2 and temper == 1.0:
            np.testing.assert_allclose(loss.item(), g["loss_" + tag], rtol=1e-4)
        noise_floor = 1e-4 * float(go["attention.2.weight"].abs().max())
        errs = {}
        for k, p in m.named_parameters():
            if go[k] is None:
                assert p.grad is None, k
                continue
            assert p.grad is not None, k
            ref, got = go[k].numpy(), p.grad.cpu().double().numpy()
            if k in ("attention.2.bias", "attention.3.bias"):  # analytically zero (softmax over T)
                assert np.abs(got - ref).max() < noise_floor, k
                continue
            err = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
            errs[k] = err
            # a pre-activation within rounding of 0 that takes the other ReLU branch moves the bias gradients behind it
            # by a whole element (test_grads_vs_oracle_small): measured on `summed` at B = 8 - layer2.convs.5.bias
            # 2.8e-2, five more tensors of the same branch 0.7 - 1.8e-2, everything else at the median.  So: every
            # tensor inside the flip-tolerant 5e-2, and the MEDIAN tensor inside 10 x the oracle's own fp32 band.
            assert err < 5e-2, "%s %s (B = %d): relative L2 grad err %.3g (oracle fp32 band %.3g)" % (tag, k, B, err, band[k])
            if B == 2 and temper == 1.0:
                np.testing.assert_allclose(p.grad.norm().item(), g["gnorm_%s_%s" % (tag, k)], rtol=5e-2)
        med, med_band = float(np.median(list(errs.values()))), float(np.median([band[k] for k in errs]))
        print("%s B = %d: median rel L2 %.3g (oracle fp32 band %.3g), worst %.3g" % (tag, B, med, med_band, max(errs.values())))
        assert med <= max(tol / 10.0, 10.0 * med_band), (tag, B, med, med_band)
        np.testing.assert_allclose(lossm.center.grad.cpu().numpy(), gco.numpy(), rtol=0, atol=tol * float(gco.abs().max()))
    for dt in ("bf16", "bf16c"):
        with pytest.raises(_hip.AirError, match="fp32"):
            m.set_compute_dtype(dt)(x.cuda())
    with pytest.raises(ValueError, match="Undefined encoder"):
        Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60, encoder_type="XYZ")
