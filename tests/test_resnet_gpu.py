"""GPU parity of the drop-in ResNet + OC-Softmax train path against the oracle and the
reference's golden vectors."""
import numpy as np
import pytest
import torch

from oracle import resnet as o_resnet
from oracle import train as o_train
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat

from _budget import check_relu_flips, conv_path, record, tol  # noqa: E402

pytestmark = pytest.mark.gpu

# test_grads_vs_oracle_small, against the fp64 oracle evaluated with the HIP run's own ReLU decisions: 10x the
# worst per-tensor relative L2 `strict` measures against the plain oracle (1.06e-5, gpurun_out/parity_measured.jsonl)
MASKED_REL_L2 = 1.1e-4
MASKED_MAX_ENTRY = 1e-3        # worst single entry of a gradient tensor, relative to the tensor's largest
FLIP_MAX_PREACT = 2e-5         # |BatchNorm output| (O(1) scale) of a pre-activation whose ReLU decision differs (measured 2.3e-6)
FLIP_MAX_COUNT = {"strict": 8, "default": 32}  # of 1.80 M ReLU inputs at (B, T) = (2, 96); measured 0 / 3


def att_T(T):
    for _ in range(3):
        T = (T + 2 - 3) // 2 + 1
    return T


def make_model():
    from asvspoof2021_air_amd.resnet import ResNet
    m = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(m)
    return m.cuda()


def test_state_dict_surface():
    from asvspoof2021_air_amd.resnet import ResNet
    m = ResNet(3, 256, resnet_type="18", nclasses=2)
    want = o_resnet.resnet18_shapes()
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got.items()) == [(k, tuple(v)) for k, v in want.items()]
    assert sum(p.numel() for p in m.parameters()) == 12450290


@pytest.mark.parametrize("path", ["strict", "default"])
@pytest.mark.parametrize("tag,B,T", [("small", 2, 96), ("full", 2, 750)])
def test_forward_vs_golden(golden, tag, B, T, path):
    with conv_path(path):
        _forward_vs_golden(golden, tag, B, T, path)


def _forward_vs_golden(golden, tag, B, T, path):
    g = golden("resnet.npz")
    m = make_model()
    x = synth_feat((B, 1, 60, T), seed=200 + T)
    for mode in ("train", "eval"):
        fill_module_(m)
        m.train(mode == "train")
        torch.manual_seed(1234)
        m.set_attention_noise(1e-5 * torch.randn(B, att_T(T), 256))
        with torch.no_grad():
            feat, mu = m(x.cuda())
        # |feat| ~ 0.5; the oracle itself matches the reference to 2e-7 here
        np.testing.assert_allclose(feat.cpu().numpy(), g["feat_%s_%s" % (tag, mode)], atol=2e-5)
        np.testing.assert_allclose(mu.cpu().numpy(), g["mu_%s_%s" % (tag, mode)], atol=2e-5)
        if mode == "train":
            sd = m.state_dict()
            for k in ("bn1.running_mean", "bn1.running_var", "layer4.1.bn2.running_mean", "bn5.running_var"):
                np.testing.assert_allclose(sd[k].cpu().numpy(), g["%s_%s" % (k, tag)], atol=tol("running_stat_atol", path))
            assert int(sd["bn1.num_batches_tracked"]) == 1


def test_transposed_view_input():
    """main_train.py:338 hands the model a non-contiguous transposed view."""
    m = make_model().eval()
    m.set_attention_noise(None)
    x = synth_feat((2, 1, 96, 60), seed=9).cuda()
    with torch.no_grad():
        a, _ = m(x.transpose(2, 3))
        b, _ = m(x.transpose(2, 3).contiguous())
    assert torch.equal(a, b)


@pytest.mark.parametrize("path", ["strict", "default"])
def test_grads_vs_oracle_small(golden, path):
    """``strict``: round 1's kernels (NO_WINO4 = 1: F(2x2,3x3) + direct), round-1 constants.  ``default``: Winograd
    kernels, the centre-gradient floor scaled by the emulated rounding ratio (tests/_budget.py)."""
    with conv_path(path):
        _grads_vs_oracle_small(golden, path)


def _grads_vs_oracle_small(golden, path):
    g = golden("resnet.npz")
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    m = make_model().train()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    x = synth_feat((2, 1, 60, 96), seed=296)
    labels = torch.tensor([0, 1])
    torch.manual_seed(1234)
    noise = 1e-5 * torch.randn(2, 12, 256)
    m.set_attention_noise(noise)
    feat, mu = m(x.cuda())
    loss, neg = lossm(feat, labels.cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss_small"], rtol=2e-5)
    # fp64 oracle gradients.  Metric: relative L2 error per tensor (a pre-activation within
    # rounding of 0 may land on the other side of a ReLU in fp32 and move one channel's
    # gradient by a whole element; elementwise the gradients agree to ~1e-5), with a loose
    # bound on the worst entry.  The golden per-tensor norms pin the result to the reference.
    p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in fill_state(o_resnet.resnet18_shapes()).items()}
    tr = o_train.OracleTrainer("resnet", p64, fill_value("center", (1, 256)).double())
    lo, no, fo, go, gco, _ = tr.loss_and_grads(x.double(), labels, noise.double())
    worst = 0.0
    for k, p in m.named_parameters():
        if go[k] is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        ref = go[k].numpy()
        got = p.grad.cpu().double().numpy()
        err = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        worst = max(worst, err)
        assert err < tol("grad_rel_l2", path), "%s: relative L2 grad err %.3g" % (k, err)
        # worst single entry against the UNMASKED oracle: 5 % of the tensor's largest under `strict`; under `default`
        # single entries carry no bound of their own here - a pre-activation that lands on the other side of a ReLU
        # moves whole product terms - they are bounded below, against the oracle evaluated with THIS run's ReLU
        # decisions, where every entry of every tensor has to agree
        e_abs, scale_k = np.abs(got - ref), np.abs(ref).max()
        if path == "strict":
            assert e_abs.max() <= tol("grad_max_entry", path) * scale_k, k
        np.testing.assert_allclose(p.grad.norm().item(), g["gnorm_" + k], rtol=tol("grad_rel_l2", path))
    # ---- the ReLU-flip account.  `default` sits ~1000x farther from the fp64 oracle than
    # `strict` (1.2e-2 against 1.1e-5 worst relative L2) where the convolutions' rounding differs 4.6x.  Claim: all
    # of the excess comes from pre-activations within rounding of 0 that take the other branch of a ReLU (two
    # samples per BatchNorm: one flipped element moves a tensor by ~1/sqrt(N)).  Proof: take the sign decisions of
    # all 18 ReLUs from the HIP run (the activated tensors it saved for backward), evaluate the fp64 oracle WITH
    # those decisions (oracle/resnet.py::ReluProbe: y = x * mask, forward and backward), and every gradient tensor
    # must agree to 10x what `strict` measures - under BOTH paths; the probe counts where the oracle's own sign
    # differed and how far from 0 those pre-activations were.
    with torch.no_grad():
        _, _, S = m._forward_impl(x.cuda().contiguous(), None, save=True)
    masks = [S["blocks"][0][1] > 0]
    for blk in S["blocks"]:
        masks += [blk[5] > 0, blk[6] > 0]
    masks.append((S["a5v"] > 0).unsqueeze(2))
    probe = o_resnet.ReluProbe([mk.cpu() for mk in masks])
    trm = o_train.OracleTrainer("resnet", p64, fill_value("center", (1, 256)).double())
    trm.relu = probe
    _, _, _, gm, gcm, _ = trm.loss_and_grads(x.double(), labels, noise.double())
    n_act, n_flip = sum(int(mk.numel()) for mk in masks), sum(probe.flips)
    worst_m, worst_e = 0.0, 0.0
    for k, p in m.named_parameters():
        if gm[k] is None:
            continue
        ref, got = gm[k].numpy(), p.grad.cpu().double().numpy()
        err = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        ent = np.abs(got - ref).max() / np.abs(ref).max()
        worst_m, worst_e = max(worst_m, err), max(worst_e, ent)
        assert err <= MASKED_REL_L2, "%s: %.3g relative L2 against the oracle under this run's ReLU decisions" % (k, err)
        assert ent <= MASKED_MAX_ENTRY, (k, float(ent))
    gc_err = float(np.abs(lossm.center.grad.cpu().numpy() - gcm.numpy()).max())
    assert gc_err <= tol("g_center_atol", path)
    record("resnet_small_relu_flips[%s]" % path, {"flips": n_flip, "of": n_act, "per_relu": probe.flips,
                                                    "max_abs_preact_of_a_flip": max(probe.flip_mag),
                                                    "masked_worst_relL2": worst_m, "masked_worst_entry": worst_e,
                                                    "masked_g_center_abs": gc_err})
    print("ReLU flips %d of %d, masked worst rel L2 %.3g entry %.3g" % (n_flip, n_act, worst_m, worst_e))
    # normalised pre-activations are O(1): a flipped one was within rounding of 0, and there are few of them
    assert max(probe.flip_mag) <= FLIP_MAX_PREACT and n_flip <= FLIP_MAX_COUNT[path], (n_flip, probe.flip_mag)
    # ... and per ReLU within the emulated rounding of the convolutions in front of it (tests/_budget.py)
    record("resnet_small_relu_flip_table[%s]" % path, check_relu_flips(probe, path, "(2, 96)"))
    record("resnet_small_g_center_abs[%s]" % path, float(np.abs(lossm.center.grad.cpu().numpy() - g["g_center"]).max()))
    record("resnet_small_worst_grad_relL2[%s]" % path, float(worst))
    # (rtol = 0: the constant IS the bound - with round 1's rtol 1e-3 on entries of magnitude 1.75 it bound nothing)
    np.testing.assert_allclose(lossm.center.grad.cpu().numpy(), g["g_center"], rtol=0, atol=tol("g_center_atol", path))
    # gradients live in the flat arena (zero-copy views)
    arena = m.arena()
    assert m.conv1.weight.grad.data_ptr() == arena.grad_view("conv1.weight").data_ptr()
    print("worst relative L2 grad err", worst)


@pytest.mark.parametrize("path", ["strict", "default"])
def test_trajectory_vs_golden(golden, path):
    with conv_path(path):
        _trajectory_vs_golden(golden, path)


def _trajectory_vs_golden(golden, path):
    """3 optimisation steps (Adam on the arena + SGD on the centre) against the reference's
    losses.  Step 1 is pre-update (tight); later steps sit on Adam's sign-SGD noise floor
    (see tests/golden/make_golden.py): the first Adam updates are lr * sign(g), so rounding-level
    gradient differences flip whole updates of near-zero-gradient weights.  There the fp32 and
    fp64 oracles differ by 1e-4 and so do the direct-conv kernels (``strict``: rtol 1e-4 at step 2, the round-1
    constant); the Winograd convolutions round at a larger multiple of a layer's output scale, and step 2's
    allowance is 1e-4 times the emulated rounding ratio of the two kernels (tests/_budget.py; measured 3e-4)."""
    g = golden("trajectory.npz")
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    m = make_model()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    tr = Trainer(m, loss_module=lossm)
    xb = synth_feat((8, 1, 60, 128), seed=300).cuda()
    labels = torch.from_numpy(g["labels"]).cuda()
    losses = []
    for it in range(3):
        torch.manual_seed(500 + it)
        m.set_attention_noise(1e-5 * torch.randn(8, 16, 256))
        loss, _ = tr.step_features(xb, labels)
        losses.append(loss.item())
    np.testing.assert_allclose(losses[0], g["losses"][0], rtol=2e-5)
    record("resnet_traj_rel[%s]" % path, [abs(a / b - 1.0) for a, b in zip(losses, g["losses"].tolist())])
    np.testing.assert_allclose(losses[1], g["losses"][1], rtol=tol("step2_rtol", path))
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-3)
    sd = m.state_dict()
    assert np.abs(sd["conv1.weight"].cpu().numpy() - g["conv1_w"]).max() <= 3 * 2 * 5e-4 + 1e-6
    np.testing.assert_allclose(tr.loss.center.detach().cpu().numpy(), g["center"], atol=1e-5)
    assert int(sd["bn1.num_batches_tracked"]) == 3
    # fc_mu got no gradient under ang_iso -> untouched by Adam
    np.testing.assert_array_equal(sd["fc_mu.weight"].cpu().numpy(), fill_value("fc_mu.weight", (2, 256)).numpy())


def test_full_path_pcm_to_loss():
    """PCM -> fused LFCC (padded, transposed) -> ResNet -> OC-Softmax -> step, vs the oracle end to end."""
    from oracle import lfcc as o_lfcc, pad as o_pad
    from oracle.filler import synth_pcm
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    m = make_model()
    m.set_attention_noise(None)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    tr = Trainer(m, loss_module=lossm, feat_len=128)
    pcm = synth_pcm(4, 9600, seed=42)  # 61 frames -> repeat-padded to 128
    labels = torch.tensor([0, 1, 1, 0])
    loss, neg = tr.step(pcm.cuda(), labels.cuda())
    feat = torch.from_numpy(o_lfcc.lfcc_forward(pcm.numpy().copy()))
    xin = torch.stack([o_pad.repeat_pad(feat[b:b + 1], 128) for b in range(4)])  # (4,1,128,60)
    otr = o_train.OracleTrainer("resnet", fill_state(o_resnet.resnet18_shapes()), fill_value("center", (1, 256)))
    lo, no, _, _, _ = otr.step(o_pad.to_model_input(xin).contiguous(), labels, None)
    np.testing.assert_allclose(loss.item(), lo.item(), rtol=1e-4)
    np.testing.assert_allclose(neg.cpu().numpy(), no.numpy(), atol=1e-4)


def test_changing_batch_geometry_never_reuses_stale_weight_buffers():
    """The weight transforms are enqueued at the start of a forward for the geometry the PREVIOUS forward saw.  A
    direct kernel's slab layout depends on the batch size and width through its tile heuristic (the short last batch
    of an epoch, resnet.py's variable feat_len): a step whose geometry differs must transform in place, not walk the
    earlier geometry's buffers.  Gradients of every step equal, bit for bit, those of a model that never prepacks."""
    a, b = make_model(), make_model()
    b.prepack_weights = False
    for m in (a, b):
        m.train()
        m.set_attention_noise(None)
    # (layer2.0's stride-2 data gradient runs 32-channel tiles at (16, 200) and 64-channel tiles at (8, 200): same buffer
    # size, another layout)
    for step, (B, T) in enumerate([(16, 200), (16, 200), (8, 200), (16, 200), (3, 96), (16, 200)]):
        x = synth_feat((B, 1, 60, T), 50 + step).cuda()
        grads = []
        for m in (a, b):
            m.zero_grad(set_to_none=True)
            feat, mu = m(x)
            (feat.sum() + 2.0 * mu.sum()).backward()
            grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
        for n in grads[0]:
            assert torch.equal(grads[0][n], grads[1][n]), (step, B, T, n)


def test_whole_module_pickle_roundtrip(tmp_path):
    """main_train.py:675-704 saves whole modules with torch.save and generate_score.py:46-48 loads them:
    after a training step (arenas bound, side stream created) the module still pickles, and the loaded copy
    scores identically."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    m = make_model()
    m.set_attention_noise(None)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    tr = Trainer(m, loss_module=lossm, feat_len=96)
    x = synth_feat((4, 1, 60, 96), seed=8).cuda()
    tr.step_features(x, torch.tensor([0, 1, 1, 0]).cuda())
    torch.save(tr.model, tmp_path / "anti-spoofing_cqcc_model.pt")
    torch.save(tr.loss, tmp_path / "anti-spoofing_loss_model.pt")
    m2 = torch.load(tmp_path / "anti-spoofing_cqcc_model.pt", weights_only=False)
    l2 = torch.load(tmp_path / "anti-spoofing_loss_model.pt", weights_only=False)
    assert list(m2.state_dict().keys()) == list(tr.model.state_dict().keys())
    for k, v in tr.model.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    tr.model.eval()
    m2.eval()
    m2.set_attention_noise(None)
    with torch.no_grad():
        f1, _ = tr.model(x)
        f2, _ = m2(x)
        s1 = tr.loss(f1, torch.zeros(4, dtype=torch.long).cuda())[1]
        s2 = l2(f2, torch.zeros(4, dtype=torch.long).cuda())[1]
    assert torch.equal(f1, f2) and torch.equal(s1, s2)
    # and the loaded module trains (arena rebuilt on first use)
    tr2 = Trainer(m2, loss_module=l2, feat_len=96)
    tr2.step_features(x, torch.tensor([0, 1, 1, 0]).cuda())


def test_optimizer_skips_without_gradients_and_rejects_frozen():
    """torch.optim.Adam semantics on the arena: a step() with no backward since zero_grad() changes nothing
    (the arena still holds the previous sums - applying them again would be a second step), frozen parameters
    are refused, and the tail (fc_mu.*) keeps its own step count."""
    from asvspoof2021_air_amd.optim import FusedAdam
    m = make_model()
    opt = FusedAdam(m, lr=5e-4)
    xb = synth_feat((4, 1, 60, 96), seed=9).cuda()
    m.set_attention_noise(None)
    feat, mu = m(xb)
    (feat.square().mean() + mu.square().mean()).backward()  # the CE-style branch: fc_mu gets gradients too
    opt.step()
    w1 = m.arena().flat.clone()
    assert opt.step_count == 1 and opt.tail_steps == 1
    opt.zero_grad()
    opt.step()  # nothing to apply
    assert torch.equal(m.arena().flat, w1) and opt.step_count == 1
    feat, _ = m(xb)
    feat.square().mean().backward()  # ang_iso-style: no gradient for fc_mu
    fc_mu = m.fc_mu.weight.detach().clone()
    opt.step()
    assert opt.step_count == 2 and opt.tail_steps == 1 and torch.equal(m.fc_mu.weight.detach(), fc_mu)
    assert not torch.equal(m.arena().flat, w1)
    m.conv1.weight.requires_grad_(False)
    with pytest.raises(RuntimeError, match="frozen"):
        opt.step()


@pytest.mark.parametrize("stride,cin,cout", [(1, 64, 64), (2, 64, 128)])
@pytest.mark.parametrize("training", [True, False])
def test_preact_block_standalone_forward(stride, cin, cout, training):
    """PreActBlock.forward on its own (resnet.py:63-69) against the oracle's block, train and eval mode."""
    from asvspoof2021_air_amd.resnet import PreActBlock
    blk = PreActBlock(cin, cout, stride)
    fill_module_(blk)
    sd = {"b." + k: v.clone() for k, v in blk.state_dict().items()}
    blk = blk.cuda().train(training)
    x = synth_feat((3, cin, 18, 40), seed=61)
    with torch.no_grad():
        got = blk(x.cuda())
    upd = {}
    want = o_resnet.preact_block(x, sd, "b", stride, training, upd)
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    if training:
        np.testing.assert_allclose(blk.bn1.running_mean.cpu().numpy(), upd["b.bn1.running_mean"].numpy(), atol=1e-6)
        with pytest.raises(NotImplementedError):
            blk(x.cuda())  # recording a graph through a lone block is not on the hot path


def test_long_utterance_beyond_lds_pooling():
    """resnet.py:23-46 pools any length; beyond T' = 148 pooled frames (about 1190 input frames) the map no longer
    fits the pooling kernel's LDS and its in-place variant takes over: forward against the oracle, and a train step
    runs (12 s utterance = 1201 frames -> T' = 151)."""
    m = make_model().eval()
    m.set_attention_noise(None)
    x = synth_feat((2, 1, 60, 1201), seed=77)
    with torch.no_grad():
        feat, _ = m(x.cuda())
    fo, _ = o_resnet.resnet18_forward(fill_state(o_resnet.resnet18_shapes()), x, training=False, noise=None)
    assert float((feat.cpu() - fo).abs().max()) <= 2e-5 * float(fo.abs().max()) + 2e-5
    m.train()
    feat, _ = m(x.cuda())
    feat.square().mean().backward()
    assert torch.isfinite(m.conv1.weight.grad).all() and float(m.conv1.weight.grad.abs().max()) > 0


def _graph_trainer(graph, seed=4242):
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    m = make_model()
    m._noise_seed = seed  # device noise ON (resnet.py:38): the replay has to draw what the eager step draws
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    tr = Trainer(m, loss_module=lossm, feat_len=128)
    if graph:
        if graph == "segments":
            tr.segment_bytes = 8 << 20
        tr.enable_graph(segments=(graph == "segments"))
    return m, tr


def test_randn_ctr_walks_the_host_offset_sequence():
    """air_randn_ctr (device-side offset, advanced by the draw) == air_randn at the offsets the host would have passed."""
    from asvspoof2021_air_amd import ops
    dev = torch.device("cuda")
    ctr = torch.tensor([17], dtype=torch.int64, device=dev)
    off = 17
    for n in (5, 4096, 1023):
        a = ops.randn_ctr((n,), dev, 99, ctr, 1e-5)
        b = ops.randn((n,), dev, 99, off, 1e-5)
        off += (n + 3) // 4
        assert torch.equal(a, b) and int(ctr.item()) == off


def test_graphed_train_step_equals_eager():
    """Trainer.enable_graph() on the ResNet - LFCC + forward (with the per-call attention noise of
    resnet.py:38 drawn from a device-side Philox offset) + OC-Softmax + backward replayed as one hipGraph chain, the
    optimisers outside - ends on bit-identical weights, centre, BatchNorm statistics and noise offset as the eager
    launches of the same one-chain step, over six steps with changing batches and a learning-rate change."""
    from oracle.filler import synth_pcm
    batches = [(synth_pcm(4, 16000, seed=500 + i).cuda(), ((torch.arange(4) + i) % 3 != 0).long().cuda()) for i in range(6)]
    ends = []
    # (round 6) "segments": the same step captured as several hipGraphs cut at backward's bucket boundaries (the form that
    # lets world > 1 launch each bucket's all-reduce between two replays) - forward / backward called directly instead
    # of through autograd: the same kernels in the same order
    for graph in (False, True, "segments"):
        m, tr = _graph_trainer(graph)
        if not graph:
            m.overlap_wgrad = False  # the capture is one chain; same launches eagerly
        losses = []
        for i, (pcm, lab) in enumerate(batches):
            if i == 3:
                tr.set_epoch(4, lr_decay=0.5, interval=4)
            losses.append(tr.step(pcm, lab)[0].item())
        torch.cuda.synchronize()
        assert (tr._graph is not None) == bool(graph)
        if graph == "segments":
            assert len(tr._graph["segments"]) >= 3
        ends.append((losses, m.arena().flat.clone(), tr.loss.center.detach().clone(), m.bn1.running_var.clone(),
                     m.layer4[1].bn2.running_mean.clone(), int(m.bn1.num_batches_tracked), int(m._noise_ctr.item())))
    for other in ends[1:]:
        (l0, w0, c0, rv0, rm0, n0, k0), (l1, w1, c1, rv1, rm1, n1, k1) = ends[0], other
        assert l0 == l1 and torch.equal(w0, w1) and torch.equal(c0, c1) and torch.equal(rv0, rv1) and torch.equal(rm0, rm1)
        assert n0 == n1 == 6 and k0 == k1 > 0


def test_graphed_noise_is_fresh_per_replay():
    """Two replays on the SAME batch from the same weights must not see the same attention noise: the saved noise
    tensor of the replay changes, and it is the draw at the advanced offset."""
    from asvspoof2021_air_amd import ops
    from oracle.filler import synth_pcm
    m, tr = _graph_trainer(True)
    m.noise_scale = 1.0  # make the draw visible in the loss
    pcm, lab = synth_pcm(4, 16000, seed=7).cuda(), torch.tensor([0, 1, 1, 0]).cuda()
    for _ in range(3):
        tr.step(pcm, lab)
    assert tr._graph is not None
    k0 = int(m._noise_ctr.item())
    flat = m.arena().flat.clone()
    l1 = tr.step(pcm, lab)[0].item()
    m.arena().flat.copy_(flat)
    l2 = tr.step(pcm, lab)[0].item()
    k2 = int(m._noise_ctr.item())
    Tp = att_T(128)
    assert k2 - k0 == 2 * ((4 * Tp * 256 + 3) // 4)
    assert l1 != l2


def test_graphed_steps_interleaved_with_eager():
    """An eager step, an external zero_grad(), a score() call or an outgrown scratch buffer between replays: same
    sequence eager-only (one chain) and with the graph enabled -> bit-identical ends."""
    from oracle.filler import synth_pcm
    batches = [(synth_pcm(4, 16000, seed=600 + i).cuda(), ((torch.arange(4) + i) % 3 != 0).long().cuda()) for i in range(10)]
    big = (synth_pcm(12, 16000, seed=78).cuda(), (torch.arange(12) % 4 != 0).long().cuda())
    ends = []
    for graph in (False, True):
        m, tr = _graph_trainer(graph)
        if not graph:
            m.overlap_wgrad = False
        losses, kept = [], None
        for i, (pcm, lab) in enumerate(batches):
            if i == 3:
                out = tr.step_features(tr.features(pcm), lab)
            elif i == 5:
                tr.feat_optimizer.zero_grad(); tr.loss_optimizer.zero_grad()
                tr.score(pcm)
                out = tr.step(pcm, lab)
            elif i == 6:
                out = tr.step_features(tr.features(big[0]), big[1])
            else:
                out = tr.step(pcm, lab)
            if i == 4:
                kept = (out[1], out[1].clone())
            losses.append(out[0].item())
        torch.cuda.synchronize()
        assert torch.equal(kept[0], kept[1])
        ends.append((losses, m.arena().flat.clone(), tr.loss.center.detach().clone(), m.bn1.running_var.clone(),
                     int(m._noise_ctr.item())))
        if graph:
            assert tr._graph is not None and tr.model.training
    (l0, w0, c0, rv0, k0), (l1, w1, c1, rv1, k1) = ends
    assert l0 == l1 and torch.equal(w0, w1) and torch.equal(c0, c1) and torch.equal(rv0, rv1) and k0 == k1
