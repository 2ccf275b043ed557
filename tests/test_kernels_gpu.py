"""GPU parity of every hand-written kernel (through the C-ABI wrappers in
asvspoof2021_air_amd.ops) against a PyTorch-CPU fp64 evaluation of the same op
(these are floating-point kernels: SURVEY.md §8c / task ③ keep a torch
reference for them).  Tolerances are relative to the output scale: the f32
MFMA is an exact fmaf chain, so the only difference is summation order."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loss as o_loss
from oracle import resnet as o_resnet
from oracle import train as o_train
from oracle.filler import synth_feat

from _budget import STRICT, conv_budget, conv_path, record, wino_conv_bound  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from asvspoof2021_air_amd import ops
    return ops


def close(got, want, rtol=2e-5, name=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, "%s: max err %.3g of scale %.3g (rel %.3g > %.3g)" % (
        name, np.abs(got - want).max(), scale, err, rtol)


CONVS = [
    # (B, Cin, H, W, Cout, k, stride, pad)
    (2, 16, 18, 75, 64, (3, 3), 1, (1, 1)),     # layer1.0.conv1
    (2, 64, 6, 70, 64, (3, 3), 1, (1, 1)),      # layer1 3x3
    (3, 64, 18, 33, 128, (3, 3), 2, (1, 1)),    # layer2.0.conv1 (stride 2, even W... odd Wo)
    (2, 64, 9, 75, 128, (3, 3), 2, (1, 1)),     # odd H and W
    (2, 128, 9, 40, 128, (3, 3), 1, (1, 1)),
    (2, 256, 5, 47, 256, (3, 3), 1, (1, 1)),
    (1, 512, 3, 94, 512, (3, 3), 1, (1, 1)),    # layer4 3x3
    (2, 16, 18, 75, 64, (1, 1), 1, (0, 0)),     # layer1.0.shortcut
    (2, 16, 18, 76, 64, (1, 1), 1, (0, 0)),     # the same with H W % 4 == 0: the streaming weight-gradient kernel
    (5, 16, 18, 750, 64, (1, 1), 1, (0, 0)),    # at the full width (ragged last pixel chunk)
    (2, 64, 18, 75, 128, (1, 1), 2, (0, 0)),    # layer2.0.shortcut
    (2, 256, 5, 47, 512, (1, 1), 2, (0, 0)),    # layer4.0.shortcut-like
    (2, 512, 3, 94, 256, (3, 3), 1, (0, 1)),    # conv5
    (3, 512, 3, 47, 256, (3, 3), 1, (0, 1)),    # conv5 geometry, odd width (dgrad = three 1x3 row convs)
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("fused", [False, True])
def test_conv2d_fwd(ops, cfg, fused):
    B, Cin, H, W, Cout, k, s, p = cfg
    x = synth_feat((B, Cin, H, W), 1)
    w = synth_feat((Cout, Cin) + k, 2, scale=0.1)
    scale = shift = res = None
    xa = x.double()
    if fused:
        scale = 1.0 + 0.2 * synth_feat((Cin,), 3)
        shift = 0.3 * synth_feat((Cin,), 4)
        xa = F.relu(xa * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    want = F.conv2d(xa, w.double(), None, s, p)
    if fused:
        res = synth_feat(tuple(want.shape), 5)
        want = want + res.double()
    got = ops.conv2d_fwd(x.cuda(), w.cuda(), s, p,
                         None if scale is None else scale.cuda(), None if shift is None else shift.cuda(),
                         relu=fused, residual=None if res is None else res.cuda())
    close(got, want, name="conv2d_fwd")


S2 = [
    # (B, Cin, H, W, Cout): stride-2 3x3 layers - layer2.0 / layer3.0 / layer4.0 conv1 geometries, odd sizes, ragged tiles
    (3, 64, 18, 33, 128), (2, 64, 9, 75, 128), (2, 128, 9, 131, 256), (2, 256, 5, 47, 512), (1, 64, 18, 750, 128),
    (2, 96, 6, 40, 128),  # three 32-channel weight-gradient tiles
    (2, 64, 4, 6, 128),   # one ragged tile per row: left and right padding columns in the same tile
    (1, 64, 3, 70, 128),  # two output rows, the first padded above; 35 output columns (a 3-pixel second tile)
    (3, 64, 2, 65, 128),  # one output row; odd width: the last valid pixel reads the right padding column
]


@pytest.mark.parametrize("s2", [0, 1, 7, 31])
@pytest.mark.parametrize("cfg", S2)
@pytest.mark.parametrize("fused", [False, True])
def test_conv2d_stride2_options(ops, cfg, fused, s2):
    """Option CONV_S2: the stride-2 3x3 forward in 4- or 8-channel K chunks - either an fmaf chain held to the direct
    kernels' constant (1e-5 of the output scale) - or (bit 4, plain input only) as six bf16 products per fp32 product on
    operands split exactly into three bf16 planes (conv_bf3.hip), held to the SAME constant; the weight gradient of the
    same layers beside it (bit 16: on the split-bf16 kernel too, where Cout % 128 == 0 and Cin % 32 == 0)."""
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 1)
    w = synth_feat((Cout, Cin, 3, 3), 2, scale=0.1).double().requires_grad_(True)
    scale = shift = None
    xa = x.double()
    if fused:
        scale = 1.0 + 0.2 * synth_feat((Cin,), 3)
        shift = 0.3 * synth_feat((Cin,), 4)
        xa = F.relu(xa * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    y = F.conv2d(xa, w, None, 2, 1)
    dy = synth_feat(tuple(y.shape), 6)
    y.backward(dy.double())
    sc, sh = (None, None) if scale is None else (scale.cuda(), shift.cuda())
    with _hip.options(CONV_S2=s2):
        got = ops.conv2d_fwd(x.cuda(), w.detach().float().cuda(), 2, 1, sc, sh, relu=fused)
        gw = ops.conv2d_wgrad(x.cuda(), dy.cuda(), tuple(w.shape), 2, 1, sc, sh, relu=fused)
    close(got, y.detach(), rtol=1e-5, name="conv2d_fwd stride 2")
    close(gw, w.grad, rtol=1e-5, name="conv2d_wgrad stride 2")


@pytest.mark.parametrize("cfg", S2 + [(2, 16, 2, 2, 64), (1, 32, 7, 61, 64), (5, 48, 11, 200, 192), (64, 256, 5, 188, 512)])
def test_conv2d_stride2_split_bf16(ops, cfg):
    """conv_bf3.hip (built): 3x3 / stride 2 forward as hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid
    on v_mfma_f32_32x32x16_bf16 with x = hi + mid + lo exact.  fp32-EQUIVALENT arithmetic or it does not ship: held to
    the direct f32 kernels' constant (STRICT conv_rtol = 1e-5 of the output scale; measured 8e-7 .. 1.5e-6, the f32 chain
    4e-7 .. 5e-7) against fp64; the layer takes this path (the prepacked planes are 6 bytes per weight); prepacked ==
    packed in place, bit for bit; shapes: every ResNet geometry, the smallest image, odd widths, 3-tile rows, B = 64 at
    layer4's size."""
    import ctypes
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    g = torch.Generator().manual_seed(B * 1000 + W)
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    nref = min(B, 3)
    want = F.conv2d(x[:nref].double(), w.double(), None, 2, 1)
    with _hip.options(CONV_S2=7):
        d = ops._conv_desc(x.shape, w.shape, 2, 1)
        assert int(_hip.lib().air_conv2d_prepack_bytes(ctypes.byref(d), 0)) == (Cout // 64) * (Cin // 16) * 9 * 2 * 3 * 1024
        got = ops.conv2d_fwd(x.cuda(), w.cuda(), 2, 1)
        pk = ops.conv2d_prepack(w.cuda(), x.shape, 2, 1, 0)
        got2 = ops.conv2d_fwd(x.cuda(), torch.full_like(w, float("nan")).cuda(), 2, 1, w_packed=pk)
    assert torch.equal(got, got2)
    close(got[:nref], want, rtol=STRICT["conv_rtol"], name="split-bf16 stride-2 forward")
    with _hip.options(CONV_S2=3):
        f32 = ops.conv2d_fwd(x.cuda(), w.cuda(), 2, 1)
    # against the f32 kernel on the whole batch (both within 1e-5 of the truth)
    assert float((got - f32).abs().max()) <= 2e-5 * float(f32.abs().max())
    e = float((got[:nref].cpu().double() - want).abs().max() / want.abs().max())
    record("conv_s2_bf3_err[%s]" % (cfg,), e)
    # the block's 1x1 / stride 2 shortcut in the same launch (air_conv2d_fwd_s2_pair): both outputs against fp64, the
    # 3x3 output bit-identical to the launch without the shortcut where the tiling is the same (two pixel tiles per wave)
    wsc = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5
    want_sc = F.conv2d(x[:nref].double(), wsc.double(), None, 2, 0)
    with _hip.options(CONV_S2=7):
        assert ops.conv2d_fwd_s2_pair_ok(w.shape, x.shape)
        y2, ysc = ops.conv2d_fwd_s2_pair(x.cuda(), w.cuda(), wsc.cuda())
        pkp = ops.conv2d_fwd_s2_pair_prepack(w.cuda(), wsc.cuda(), x.shape)
        nan = torch.full_like(w, float("nan")).cuda()
        y3, ysc3 = ops.conv2d_fwd_s2_pair(x.cuda(), nan, nan[:, :, :1, :1].contiguous(), packed=pkp)
    assert torch.equal(y2, y3) and torch.equal(ysc, ysc3)
    close(y2[:nref], want, rtol=STRICT["conv_rtol"], name="pair forward: 3x3")
    close(ysc[:nref], want_sc, rtol=STRICT["conv_rtol"], name="pair forward: shortcut")
    with _hip.options(CONV_S2=3):
        assert not ops.conv2d_fwd_s2_pair_ok(w.shape, x.shape) and ops.conv2d_fwd_s2_pair(x.cuda(), w.cuda(), wsc.cuda()) is None


@pytest.mark.parametrize("cfg", S2 + [(2, 40, 7, 66, 64), (1, 24, 5, 9, 64)])  # + input-channel counts that fill no 32-channel tile
@pytest.mark.parametrize("one_pass", [True, False, "bf3"])
def test_conv2d_stride2_dgrad_one_pass(ops, cfg, one_pass):
    """Option CONV_S2 bit 2: the 3x3 / stride 2 data gradient with all four parity classes in one pass over dy
    (conv_s2_dgrad_kernel) or as four class launches; alone, with `accumulate`, and joined with the 1x1 / stride 2
    shortcut's data gradient (air_conv2d_dgrad_s2_pair, resnet.py:56-66) - each against the fp64 autograd of
    F.conv2d, held to the direct kernels' constant (1e-5 of the output scale).  Prepacked weights: bit-identical."""
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 1).double().requires_grad_(True)
    w = synth_feat((Cout, Cin, 3, 3), 2, scale=0.1)
    wsc = synth_feat((Cout, Cin, 1, 1), 3, scale=0.2)
    y = F.conv2d(x, w.double(), None, 2, 1)
    ysc = F.conv2d(x, wsc.double(), None, 2, 0)
    assert y.shape == ysc.shape
    dy, dysc = synth_feat(tuple(y.shape), 6), synth_feat(tuple(y.shape), 8)
    g3, = torch.autograd.grad(y, x, dy.double())
    gsc, = torch.autograd.grad(ysc, x, dysc.double())
    acc = synth_feat((B, Cin, H, W), 7)
    xs = (B, Cin, H, W)
    # "bf3": option bit 8 - the paired data gradient as six bf16 products per fp32 product (conv_bf3.hip) where the layer
    # fits it (Cin % 64 == 0, Cout % 16 == 0; the one-pass f32 kernel otherwise), held to the SAME 1e-5
    with _hip.options(CONV_S2=15 if one_pass == "bf3" else (3 if one_pass else 1)):
        assert ops.conv2d_dgrad_s2_pair_ok(w.shape, xs) == bool(one_pass)
        if one_pass == "bf3" and Cin % 64 == 0 and Cout % 16 == 0:
            import ctypes
            d = ops._conv_desc(xs, w.shape, 2, 1)
            assert int(_hip.lib().air_conv2d_dgrad_s2_pair_prepack_bytes(ctypes.byref(d))) == (Cin // 64) * (Cout // 16) * 10 * 6 * 1024
        got = ops.conv2d_dgrad(dy.cuda(), w.cuda(), xs, 2, 1)
        close(got, g3, rtol=1e-5, name="stride-2 dgrad")
        got = ops.conv2d_dgrad(dy.cuda(), w.cuda(), xs, 2, 1, accumulate=acc.cuda())
        close(got, g3 + acc.double(), rtol=1e-5, name="stride-2 dgrad + accumulate")
        pd = ops.conv2d_prepack(w.cuda(), xs, 2, 1, 1)
        assert torch.equal(ops.conv2d_dgrad(dy.cuda(), torch.full_like(w, float("nan")).cuda(), xs, 2, 1,
                                            accumulate=acc.cuda(), w_packed=pd), got)
        pair = ops.conv2d_dgrad_s2_pair(dy.cuda(), w.cuda(), dysc.cuda(), wsc.cuda(), xs)
        if not one_pass:
            assert pair is None and ops.conv2d_dgrad_s2_pair_prepack(w.cuda(), wsc.cuda(), xs) is None
            return
        close(pair, g3 + gsc, rtol=1e-5, name="stride-2 pair dgrad")
        inplace = acc.cuda().clone()  # accumulate aliasing dx
        ops.conv2d_dgrad_s2_pair(dy.cuda(), w.cuda(), dysc.cuda(), wsc.cuda(), xs, accumulate=inplace, out=inplace)
        close(inplace, g3 + gsc + acc.double(), rtol=1e-5, name="stride-2 pair dgrad + accumulate")
        pk = ops.conv2d_dgrad_s2_pair_prepack(w.cuda(), wsc.cuda(), xs)
        nan = torch.full_like(w, float("nan")).cuda()
        assert torch.equal(ops.conv2d_dgrad_s2_pair(dy.cuda(), nan, dysc.cuda(), nan[:, :, :1, :1].contiguous(), xs, packed=pk), pair)
        with pytest.raises(_hip.AirError):
            ops.conv2d_dgrad_s2_pair(dy.cuda(), w.cuda(), dysc.cuda(), wsc.cuda(), xs, packed=pk[:pk.numel() // 2].clone())


@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d_prepack(ops, cfg):
    """air_conv2d_prepack for every layer kind (Winograd transforms or the direct kernels' slabs - one or several per
    pass): forward and data gradient with the prepacked buffer are bit-identical to the calls that transform in place."""
    B, Cin, H, W, Cout, k, s, p = cfg
    x = synth_feat((B, Cin, H, W), 1).cuda()
    w = synth_feat((Cout, Cin) + k, 2, scale=0.1).cuda()
    y = ops.conv2d_fwd(x, w, s, p)
    dy = synth_feat(tuple(y.shape), 6).cuda()
    acc = synth_feat((B, Cin, H, W), 7).cuda()
    dx = ops.conv2d_dgrad(dy, w, x.shape, s, p, accumulate=acc)
    pf = ops.conv2d_prepack(w, x.shape, s, p, 0)
    pd = ops.conv2d_prepack(w, x.shape, s, p, 1)
    assert pf is not None and pd is not None
    w_gone = torch.full_like(w, float("nan"))  # the packed buffers carry everything the kernels read
    y2 = ops.conv2d_fwd(x, w_gone, s, p, w_packed=pf)
    dx2 = ops.conv2d_dgrad(dy, w_gone, x.shape, s, p, accumulate=acc, w_packed=pd)
    assert torch.equal(y, y2)
    assert torch.equal(dx, dx2)
    # a buffer smaller than what the layer consumes (e.g. prepacked under other dispatch options) is refused, not walked
    from asvspoof2021_air_amd import _hip
    with pytest.raises(_hip.AirError):
        ops.conv2d_fwd(x, w, s, p, w_packed=pf[:pf.numel() // 2].clone())


@pytest.mark.parametrize("cfg", [(2, 64, 18, 75, 128), (3, 128, 9, 94, 256)])
def test_conv2d_prepack_layout_mismatch(ops, cfg):
    """air_conv2d_prepack(pass 0) writes split-bf16 planes (6 bytes per weight) for a bf3-eligible
    3x3 / stride 2 layer, but a forward WITH a BatchNorm + ReLU prologue or a residual takes the f32 kernel: it must not
    read those planes as f32 slabs.  (a) such calls with the prepacked buffer equal the calls without it bit for bit;
    (b) a buffer packed under other dispatch options (CONV_S2 bit 4 flipped in between) is refused by layout, not size."""
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 1).cuda()
    w = synth_feat((Cout, Cin, 3, 3), 2, scale=0.1).cuda()
    scale = (1.0 + 0.2 * synth_feat((Cin,), 3)).cuda()
    shift = (0.3 * synth_feat((Cin,), 4)).cuda()
    with _hip.options(CONV_S2=31):
        pk = ops.conv2d_prepack(w, x.shape, 2, 1, 0)
        assert pk._air_pack_layout == 3  # AIR_PACK_BF3
        y0 = ops.conv2d_fwd(x, w, 2, 1, scale, shift, relu=True)
        y1 = ops.conv2d_fwd(x, w, 2, 1, scale, shift, relu=True, w_packed=pk)
        assert torch.equal(y0, y1)
        res = synth_feat(tuple(y0.shape), 8).cuda()
        r0 = ops.conv2d_fwd(x, w, 2, 1, residual=res)
        r1 = ops.conv2d_fwd(x, w, 2, 1, residual=res, w_packed=pk)
        assert torch.equal(r0, r1)
        want = F.conv2d(F.relu(x.double().cpu() * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)),
                        w.double().cpu(), None, 2, 1)
        close(y1, want, rtol=1e-5, name="prologue forward beside a bf3 prepack buffer")
    with _hip.options(CONV_S2=3):  # bit 4 off: the layer now consumes f32 slabs
        with pytest.raises(_hip.AirError):
            ops.conv2d_fwd(x, w, 2, 1, w_packed=pk)
        pk32 = ops.conv2d_prepack(w, x.shape, 2, 1, 0)
        assert pk32._air_pack_layout == 1
    with _hip.options(CONV_S2=31):
        with pytest.raises(_hip.AirError):
            ops.conv2d_fwd(x, w, 2, 1, w_packed=pk32)


@pytest.mark.parametrize("cfg", CONVS)
def test_conv2d_dgrad(ops, cfg):
    B, Cin, H, W, Cout, k, s, p = cfg
    x = synth_feat((B, Cin, H, W), 1).double().requires_grad_(True)
    w = synth_feat((Cout, Cin) + k, 2, scale=0.1)
    y = F.conv2d(x, w.double(), None, s, p)
    dy = synth_feat(tuple(y.shape), 6)
    y.backward(dy.double())
    got = ops.conv2d_dgrad(dy.cuda(), w.cuda(), (B, Cin, H, W), s, p)
    close(got, x.grad, name="conv2d_dgrad")
    acc = synth_feat((B, Cin, H, W), 7)
    got2 = ops.conv2d_dgrad(dy.cuda(), w.cuda(), (B, Cin, H, W), s, p, accumulate=acc.cuda())
    close(got2, x.grad + acc.double(), name="conv2d_dgrad+acc")


def test_conv2d_prepack_batch(ops):
    """air_conv2d_prepack_begin / _flush: the weight transforms of many layers recorded and run as one launch per 32
    (Winograd transforms and direct slabs apart) - every buffer bit-identical to the layer's own prepack launch; more
    than 32 jobs of a kind (the table of one launch) flush on the way; a data gradient that would have to pack in place
    inside the block is refused instead of running on weights that are not there yet."""
    from asvspoof2021_air_amd import _hip
    jobs = []
    for rep in range(3):
        for B, Cin, H, W, Cout, k, s, p in CONVS:
            w = synth_feat((Cout, Cin) + k, 20 + rep, scale=0.1).cuda()
            for which in (0, 1):
                jobs.append((w, (B, Cin, H, W), s, p, which))
    sizes = [ops.conv2d_prepack(w, xs, s, p, which) for w, xs, s, p, which in jobs]
    assert sum(a is not None for a in sizes) > 64
    # (into zeroed buffers: a layout's alignment gaps are never written)
    zeros = lambda a: None if a is None else torch.zeros_like(a)
    alone = [ops.conv2d_prepack(w, xs, s, p, which, out=zeros(a)) for (w, xs, s, p, which), a in zip(jobs, sizes)]
    wsc = synth_feat((128, 64, 1, 1), 5, scale=0.2).cuda()
    w3 = synth_feat((128, 64, 3, 3), 6, scale=0.1).cuda()
    pair_alone = ops.conv2d_dgrad_s2_pair_prepack(w3, wsc, (2, 64, 18, 75))
    pair_alone = ops.conv2d_dgrad_s2_pair_prepack(w3, wsc, (2, 64, 18, 75), out=torch.zeros_like(pair_alone))
    with ops.prepack_batch():
        batched = [ops.conv2d_prepack(w, xs, s, p, which, out=zeros(a)) for (w, xs, s, p, which), a in zip(jobs, sizes)]
        pair_batched = ops.conv2d_dgrad_s2_pair_prepack(w3, wsc, (2, 64, 18, 75), out=torch.zeros_like(pair_alone))
        dy = synth_feat((2, 128, 9, 38), 7).cuda()
        with pytest.raises(_hip.AirError):
            ops.conv2d_dgrad_s2_pair(dy, w3, dy, wsc, (2, 64, 18, 75))
    for a, b in zip(alone, batched):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert torch.equal(pair_alone, pair_batched)
    # outside the block everything launches at once again
    assert torch.equal(ops.conv2d_prepack(*jobs[0], out=zeros(sizes[0])), alone[0])


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("fused", [False, True])
def test_conv2d_wgrad(ops, cfg, fused):
    B, Cin, H, W, Cout, k, s, p = cfg
    x = synth_feat((B, Cin, H, W), 1)
    w = synth_feat((Cout, Cin) + k, 2, scale=0.1).double().requires_grad_(True)
    scale = shift = None
    xa = x.double()
    if fused:
        scale = 1.0 + 0.2 * synth_feat((Cin,), 3)
        shift = 0.3 * synth_feat((Cin,), 4)
        xa = F.relu(xa * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    y = F.conv2d(xa, w, None, s, p)
    dy = synth_feat(tuple(y.shape), 6)
    y.backward(dy.double())
    got = ops.conv2d_wgrad(x.cuda(), dy.cuda(), tuple(w.shape), s, p,
                           None if scale is None else scale.cuda(), None if shift is None else shift.cuda(),
                           relu=fused)
    close(got, w.grad, name="conv2d_wgrad")


def test_conv1_direct(ops):
    """ResNet conv1: 1->16, 9x3, stride (3,1), pad (1,1) (resnet.py:131)."""
    x = synth_feat((3, 1, 60, 96), 1)
    w = synth_feat((16, 1, 9, 3), 2, scale=0.3).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, (3, 1), (1, 1))
    got = ops.conv2d_fwd(x.cuda(), w.detach().float().cuda(), (3, 1), (1, 1))
    close(got, y, name="conv1 fwd")
    dy = synth_feat(tuple(y.shape), 3)
    y.backward(dy.double())
    gw = ops.conv2d_wgrad(x.cuda(), dy.cuda(), (16, 1, 9, 3), (3, 1), (1, 1))
    close(gw, w.grad, name="conv1 wgrad")


@pytest.mark.parametrize("W", [3, 8, 32, 47, 48, 49])
def test_conv1_direct_wgrad_short_rows(ops, W):
    """Short feat_len: the row-staged conv1 weight gradient folds its per-segment sums (3 x 144 x 3 floats) in the
    LDS its staged rows leave behind; for W <= 48 the sums are the larger of the two (round-2 advisor finding)."""
    x = synth_feat((2, 1, 60, W), 11)
    w = synth_feat((16, 1, 9, 3), 2, scale=0.3).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, (3, 1), (1, 1))
    dy = synth_feat(tuple(y.shape), 12)
    y.backward(dy.double())
    gw = ops.conv2d_wgrad(x.cuda(), dy.cuda(), (16, 1, 9, 3), (3, 1), (1, 1))
    close(gw, w.grad, name="conv1 wgrad W=%d" % W)


# (160, 8, 5, 12): 160 splits per channel - the finalize kernels' ordered fold walks three 64-lane chunks
@pytest.mark.parametrize("shape", [(4, 16, 18, 75), (3, 64, 9, 375), (2, 256, 1, 94), (5, 128, 33), (160, 8, 5, 12)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm(ops, shape, relu):
    C = shape[1]
    x = synth_feat(shape, 1) * 2.0 + 0.5
    gamma = 1.0 + 0.3 * synth_feat((C,), 2)
    beta = 0.2 * synth_feat((C,), 3)
    rm = 0.1 * synth_feat((C,), 4)
    rv = 1.0 + 0.1 * synth_feat((C,), 5).abs()
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(xd, rm_ref, rv_ref, gd, bd, True, 0.1, 1e-5)
    if relu:
        y = F.relu(y)
    dy = synth_feat(shape, 6)
    y.backward(dy.double())
    rmg, rvg = rm.cuda(), rv.cuda()
    xg = x.cuda()
    mean, invstd, scale, shift = ops.bn_stats(xg, gamma.cuda(), beta.cuda(), rmg, rvg)
    yg = ops.bn_apply(xg, scale, shift, relu)
    close(yg, y, name="bn fwd")
    close(rmg, rm_ref, name="running_mean")
    close(rvg, rv_ref, name="running_var")
    dx, dg, db = ops.bn_bwd(xg, dy.cuda(), mean, invstd, gamma.cuda(), beta.cuda(), relu)
    close(dx, xd.grad, rtol=1e-4, name="bn dx")
    close(dg, gd.grad, rtol=1e-4, name="bn dgamma")
    close(db, bd.grad, rtol=1e-4, name="bn dbeta")
    base = synth_feat(shape, 7)
    dx2, _, _ = ops.bn_bwd(xg, dy.cuda(), mean, invstd, gamma.cuda(), beta.cuda(), relu,
                           dx=base.cuda(), accumulate=True)
    close(dx2, xd.grad + base.double(), rtol=1e-4, name="bn dx accumulate")
    # eval-mode coefficients
    sc, sh = ops.bn_eval_coeffs(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda())
    ye = F.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    close(ops.bn_apply(xg, sc, sh, False), ye, name="bn eval")


@pytest.mark.parametrize("B,C,T,with_noise", [(3, 256, 94, True), (2, 256, 12, False), (1, 64, 7, True),
                                              (2, 256, 149, True), (2, 256, 400, False)])  # > 148: the in-place variant
def test_selfatt_pool(ops, B, C, T, with_noise):
    x = synth_feat((B, C, T), 1).abs()
    att = synth_feat((1, C), 2, scale=0.1)
    noise = 1e-3 * synth_feat((B, T, C), 3) if with_noise else None
    xd = x.double().requires_grad_(True)
    ad = att.double().requires_grad_(True)
    want = o_resnet.self_attention_pool(xd.permute(0, 2, 1), ad, None if noise is None else noise.double())
    dout = synth_feat((B, 2 * C), 4)
    want.backward(dout.double())
    xg, ag = x.cuda(), att.cuda()
    ng = None if noise is None else noise.cuda()
    out, alpha = ops.selfatt_pool_fwd(xg, ag, ng)
    close(out, want, name="selfatt fwd")
    dx, datt = ops.selfatt_pool_bwd(xg, ag, ng, alpha, out, dout.cuda())
    close(dx, xd.grad, rtol=1e-4, name="selfatt dx")
    close(datt.sum(0, keepdim=True), ad.grad, rtol=1e-4, name="selfatt datt")


def test_selfattention_module_mean_only(ops):
    """SelfAttention(hidden, mean_only=True).forward (resnet.py:12, :43-44): the attention-weighted sum alone - the first
    half of the full pooling's output, against the oracle's restatement of resnet.py:23-46 (round 5: raised)."""
    from asvspoof2021_air_amd.resnet import SelfAttention
    B, T, H = 3, 47, 256
    inp = synth_feat((B, T, H), 1).abs()
    m = SelfAttention(H, mean_only=True).cuda()
    with torch.no_grad():
        m.att_weights.copy_(synth_feat((1, H), 2, scale=0.1))
        got = m(inp.cuda())
    want = o_resnet.self_attention_pool(inp.double(), m.att_weights.detach().cpu().double(), None)[:, :H]
    assert got.shape == (B, H)
    close(got, want, name="mean_only pooling")
    full = SelfAttention(H).cuda()
    with torch.no_grad():
        full.att_weights.copy_(m.att_weights)
        assert torch.equal(full(inp.cuda())[:, :H], got)


# (the last four: K not a multiple of 64 / below 64 - the lanes of linear_fwd_kernel's lane-strided k loop run
# different trip counts, idle lanes included, the structure VERDICT r3 item 8 asked a regression test for: a build with
# `#pragma unroll 16` on that loop has a per-lane remainder loop in front of the unrolled one)
@pytest.mark.parametrize("M,K,N", [(8, 512, 256), (3, 256, 2), (5, 3072, 256), (4, 100, 7), (9, 130, 33), (17, 40, 5),
                                   (2, 1100, 3)])
def test_linear(ops, M, K, N):
    x = synth_feat((M, K), 1)
    w = synth_feat((N, K), 2, scale=0.05)
    b = synth_feat((N,), 3)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    y = F.linear(xd, wd, bd)
    dy = synth_feat((M, N), 4)
    y.backward(dy.double())
    close(ops.linear_fwd(x.cuda(), w.cuda(), b.cuda()), y, name="linear fwd")
    dx, dw, db = ops.linear_bwd(x.cuda(), w.cuda(), dy.cuda())
    close(dx, xd.grad, name="linear dx")
    close(dw, wd.grad, name="linear dw")
    close(db, bd.grad, name="linear db")


@pytest.mark.parametrize("mode", ["mixed", "all0", "all1"])
def test_ocsoftmax(ops, golden, mode):
    g = golden("ocsoftmax.npz")
    x = torch.from_numpy(g["feats_" + mode]).cuda()
    c = torch.from_numpy(g["center"]).cuda()
    lab = torch.from_numpy(g["labels_" + mode]).cuda()
    loss, neg = ops.ocsoftmax_fwd(x, c, lab, 0.9, 0.2, 20.0)
    np.testing.assert_allclose(loss.item(), g["loss_" + mode], rtol=2e-6)
    np.testing.assert_allclose(neg.cpu().numpy(), g["negscores_" + mode], atol=1e-6)
    dx, dc = ops.ocsoftmax_bwd(x, c, lab, 0.9, 0.2, 20.0)
    np.testing.assert_allclose(dx.cpu().numpy(), g["gfeat_" + mode], atol=2e-7, rtol=1e-4)
    np.testing.assert_allclose(dc.cpu().numpy(), g["gcenter_" + mode], atol=2e-6, rtol=1e-4)
    gs = torch.tensor([0.5], device="cuda")
    dx2, _ = ops.ocsoftmax_bwd(x, c, lab, 0.9, 0.2, 20.0, gscale=gs)
    np.testing.assert_allclose(dx2.cpu().numpy(), 0.5 * g["gfeat_" + mode], atol=2e-7, rtol=1e-4)
    # independent fp64 closed form
    l64, n64, gx, gc = o_loss.ocsoftmax_grads_f64(g["feats_" + mode], g["center"], g["labels_" + mode], 0.9, 0.2, 20.0)
    np.testing.assert_allclose(dx.cpu().numpy(), gx, atol=2e-7, rtol=1e-4)


@pytest.mark.parametrize("B,D", [(1, 256), (7, 100), (130, 256), (64, 300), (33, 1100), (4096, 64)])
def test_ocsoftmax_shapes(ops, B, D):
    """Ragged row / dimension counts through the 16-wave kernels (rows in groups of 4 per wave, dimension chunks of a
    power of two, row groups per dimension) against the fp64 closed form."""
    x = synth_feat((B, D), 11)
    c = synth_feat((1, D), 12)
    lab = (torch.arange(B) % 3 == 0).long()
    l64, n64, gx, gc = o_loss.ocsoftmax_grads_f64(x.numpy(), c.numpy(), lab.numpy(), 0.9, 0.2, 20.0)
    loss, neg = ops.ocsoftmax_fwd(x.cuda(), c.cuda(), lab.cuda(), 0.9, 0.2, 20.0)
    np.testing.assert_allclose(loss.item(), l64, rtol=5e-6)
    np.testing.assert_allclose(neg.cpu().numpy(), n64, atol=2e-6)
    dx, dc = ops.ocsoftmax_bwd(x.cuda(), c.cuda(), lab.cuda(), 0.9, 0.2, 20.0)
    np.testing.assert_allclose(dx.cpu().numpy(), gx, atol=2e-7 + 1e-6 / B, rtol=2e-4)
    np.testing.assert_allclose(dc.cpu().numpy().reshape(-1), np.asarray(gc).reshape(-1), atol=5e-6, rtol=2e-4)


def test_ocsoftmax_refuses_more_rows_than_its_workgroup_holds(ops):
    x = synth_feat((4097, 8), 13).cuda()
    c = synth_feat((1, 8), 14).cuda()
    lab = torch.zeros(4097, dtype=torch.long, device="cuda")
    with pytest.raises(Exception):
        ops.ocsoftmax_fwd(x, c, lab, 0.9, 0.2, 20.0)
    with pytest.raises(Exception):
        ops.ocsoftmax_bwd(x, c, lab, 0.9, 0.2, 20.0)


def test_adam_sgd(ops):
    n = 100003
    p = synth_feat((n,), 1)
    g = synth_feat((n,), 2)
    pg, m, v = p.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr, mr, vr = p.clone(), torch.zeros(n), torch.zeros(n)
    for step in range(1, 4):
        gs = g * step
        ops.adam_step(pg, gs.cuda(), m, v, step)
        o_train.adam_step_(pr, gs, mr, vr, step)
        np.testing.assert_allclose(pg.cpu().numpy(), pr.numpy(), atol=2e-7)
    c = synth_feat((1, 256), 3)
    cg = c.cuda()
    ops.sgd_step(cg, g[:256].reshape(1, 256).cuda(), 5e-4)
    np.testing.assert_allclose(cg.cpu().numpy(), (c - 5e-4 * g[:256].reshape(1, 256)).numpy(), atol=1e-7)
    ops.sgd_step(cg, g[:256].reshape(1, 256).cuda(), 5e-4, grad_scale=0.5)


# Winograd F(2x2,3x3) path (conv_wino.hip): every 3x3 / stride 1 / pad 1 forward without fused
# prologue and every such dgrad.  Shapes cover both tile-group forms (1x32, 2x16), odd H / W,
# an odd group count (padding group), W = 2 / H = 1, more work items than workgroups (each
# persistent workgroup walks several items, with and without an output-channel-tile change
# between them) and the residual epilogue.
WINO = [
    # (B, Cin, H, W, Cout)
    (2, 64, 18, 75, 64),
    (1, 512, 3, 94, 512),
    (3, 32, 5, 47, 64),
    (3, 8, 2, 64, 64),
    (1, 8, 1, 2, 64),
    (2, 8, 7, 3, 128),
    (40, 16, 18, 150, 64),
    (60, 32, 9, 75, 128),
    (5, 64, 18, 750, 64),     # wgrad: 24 stages per tile row, right edge inside a lane chunk
    (7, 128, 9, 375, 64),     # odd W: columns W and W+1 are both seen by the last tile
    (3, 64, 4, 33, 128),
]


@pytest.mark.parametrize("opts", [dict(WINO4_XCD=x, WINO4_SPLIT=sp, WINO4_TH3=th)
                                  for x in (0, 1, 2) for sp in (0, 1) for th in (0, 1, 2)])
@pytest.mark.parametrize("cfg", [(2, 64, 18, 75, 64), (6, 32, 9, 100, 128), (9, 16, 7, 130, 64), (24, 8, 6, 40, 256)])
def test_wino4_item_dealing_options(ops, cfg, opts):
    """Every way the F(4x4 | 3x4, 3x3) kernel deals its work items (whole chip / XCD-contiguous / XCD-strided),
    with and without half-cut tail items, 4- and 3-row tiles: the same result (cfg0 has a dealing group with one
    item for two workgroups - a cut item and nothing else; the others leave 1 .. 3 whole rounds plus a tail)."""
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 21)
    w = synth_feat((Cout, Cin, 3, 3), 22, scale=0.1)
    res = synth_feat((B, Cout, H, W), 23)
    want = F.conv2d(x.double(), w.double(), None, 1, 1) + res.double()
    with _hip.options(**opts):
        got = ops.conv2d_fwd(x.cuda(), w.cuda(), 1, 1, residual=res.cuda())
    close(got, want, rtol=wino_conv_bound(), name="wino4 %s" % (opts,))


@pytest.mark.parametrize("cfg", [(2, 16, 18, 75, 64), (64, 16, 18, 150, 64), (3, 48, 9, 40, 64), (24, 16, 5, 47, 128)])
def test_wino4_half_slab(ops, cfg):
    """Output channel counts 32 n + 16 on the F(3x4 | 4x4, 3x3) kernel: the last 32-channel slab runs half empty (zero
    weight rows, nothing loaded or stored for them) - the data gradient of the ResNet's 16 -> 64 layer (resnet.py:56)
    and of a 48-channel neighbour - with an accumulate operand against fp64, the tensors next to the written one
    untouched; the forward of the same layers beside it."""
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 11)
    w = synth_feat((Cout, Cin, 3, 3), 12, scale=0.1)
    res = synth_feat((B, Cout, H, W), 13)
    xd = x.double().requires_grad_(True)
    y = F.conv2d(xd, w.double(), None, 1, 1)
    dy = synth_feat((B, Cout, H, W), 14)
    y.backward(dy.double())
    acc = synth_feat((B, Cin, H, W), 15)
    wb = wino_conv_bound()
    for split in (0, 1):
        with _hip.options(WINO4_SPLIT=split):
            close(ops.conv2d_fwd(x.cuda(), w.cuda(), 1, 1, residual=res.cuda()), y + res.double(), rtol=wb,
                  name="fwd + residual")
            # dx as the middle third of a guarded buffer: a store to a channel that does not exist would land in it
            buf = torch.full((3,) + tuple(x.shape), 7.0, device="cuda")
            got = ops.conv2d_dgrad(dy.cuda(), w.cuda(), (B, Cin, H, W), 1, 1, accumulate=acc.cuda(), out=buf[1])
            close(got, xd.grad + acc.double(), rtol=wb, name="dgrad + accumulate")
            assert bool((buf[0] == 7.0).all()) and bool((buf[2] == 7.0).all())


@pytest.mark.parametrize("cfg", WINO)
def test_conv2d_winograd(ops, cfg):
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 11)
    w = synth_feat((Cout, Cin, 3, 3), 12, scale=0.1)
    res = synth_feat((B, Cout, H, W), 13)
    xd = x.double().requires_grad_(True)
    y = F.conv2d(xd, w.double(), None, 1, 1)
    got = ops.conv2d_fwd(x.cuda(), w.cuda(), 1, 1)
    # fp32 Winograd F(4x4,3x3) / F(3x4,3x3) (conv_wino4.hip; F(2x2,3x3) below W = 4): the transforms carry
    # constants up to 8 and 1/24, so it is noisier than the direct f32 MFMA chain.  Bound = the emulated error of
    # the worst ResNet layer x 1.5 (tests/_budget.py: about 1.5e-5 of the output scale; measured 2e-6 .. 1.05e-5)
    wb = wino_conv_bound()
    assert STRICT["conv_rtol"] < wb < 2e-5, wb
    close(got, y, rtol=wb, name="winograd fwd")
    got = ops.conv2d_fwd(x.cuda(), w.cuda(), 1, 1, residual=res.cuda())
    close(got, y + res.double(), rtol=wb, name="winograd fwd + residual")
    dy = synth_feat((B, Cout, H, W), 14)
    y.backward(dy.double())
    got = ops.conv2d_dgrad(dy.cuda(), w.cuda(), (B, Cin, H, W), 1, 1)
    close(got, xd.grad, rtol=wb, name="winograd dgrad")
    acc = synth_feat((B, Cin, H, W), 15)
    got = ops.conv2d_dgrad(dy.cuda(), w.cuda(), (B, Cin, H, W), 1, 1, accumulate=acc.cuda())
    close(got, xd.grad + acc.double(), rtol=wb, name="winograd dgrad + accumulate")
    # the same calls in round 1's configuration (F(2x2,3x3)) and on the direct f32-MFMA kernels hold round 1's 1e-5
    for path in ("strict", "direct"):
        with conv_path(path):
            close(ops.conv2d_fwd(x.cuda(), w.cuda(), 1, 1, residual=res.cuda()), y + res.double(),
                  rtol=STRICT["conv_rtol"], name=path + " fwd + residual")
            close(ops.conv2d_dgrad(dy.cuda(), w.cuda(), (B, Cin, H, W), 1, 1, accumulate=acc.cuda()),
                  xd.grad + acc.double(), rtol=STRICT["conv_rtol"], name=path + " dgrad + accumulate")
    # weight gradient: Winograd F(3x3,2x2) when both channel counts are multiples of 64
    wd = w.double().requires_grad_(True)
    F.conv2d(x.double(), wd, None, 1, 1).backward(dy.double())
    got = ops.conv2d_wgrad(x.cuda(), dy.cuda(), tuple(w.shape), 1, 1)
    close(got, wd.grad, rtol=1e-5, name="winograd wgrad")


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("cfg", [(2, 64, 18, 75, 64), (6, 32, 9, 100, 128), (5, 64, 5, 47, 256), (24, 16, 3, 94, 64),
                                 (3, 64, 4, 33, 128), (7, 128, 9, 375, 64)])
def test_batchnorm_statistics_from_the_conv_epilogue(ops, cfg, split):
    """Round 4: ``conv2d_fwd(..., stats=True)`` + ``bn_stats(..., stats_in=records)`` against the BatchNorm's own pass
    over the same tensor and against fp64.  Shapes: W % 4 = 3 / 0 / 3 / 2 / 1 / 3 (tiles that straddle the right edge
    contribute their valid columns only), H % 3 != 0 (rows computed for nothing are left out), 4-row tiles (H = 4),
    2 x 8 tile groups, tile groups beyond the image stack (empty records), cut tail items (the owner of the lower
    half writes the records after adding the partner's sums), with and without a residual.  The mean and the
    variance agree with the fp64 statistics of the STORED tensor to 2e-6 relative (fp32 records of <= 192 values,
    merged in fp64); running statistics are updated the same way."""
    from asvspoof2021_air_amd import _hip
    B, Cin, H, W, Cout = cfg
    x = synth_feat((B, Cin, H, W), 31).cuda()
    w = synth_feat((Cout, Cin, 3, 3), 32, scale=0.1).cuda()
    res = (synth_feat((B, Cout, H, W), 33) + 0.7).cuda()  # (a mean well away from 0: the shifted sums matter)
    gamma, beta = (1.0 + 0.3 * synth_feat((Cout,), 34)).cuda(), (0.2 * synth_feat((Cout,), 35)).cuda()
    with _hip.options(WINO4_SPLIT=split):
        for residual in (None, res):
            y, rec = ops.conv2d_fwd(x, w, 1, 1, residual=residual, stats=True)
            assert rec is not None
            assert torch.equal(y, ops.conv2d_fwd(x, w, 1, 1, residual=residual))  # the records change nothing in y
            rm0, rv0 = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
            rm1, rv1 = rm0.clone(), rv0.clone()
            a = ops.bn_stats(y, gamma, beta, rm0, rv0)
            b = ops.bn_stats(y, gamma, beta, rm1, rv1, stats_in=rec)
            yd = y.double()
            mean64 = yd.mean((0, 2, 3))
            var64 = yd.var((0, 2, 3), unbiased=False)
            n = B * H * W
            for got in (a, b):
                close(got[0], mean64, 2e-6, "mean")
                close(got[1], 1.0 / torch.sqrt(var64 + 1e-5), 2e-6, "invstd")
            close(b[2], a[2], 2e-6, "scale")
            close(b[3], a[3], 4e-6, "shift")
            close(rm1, 0.1 * mean64, 2e-6, "running mean")
            close(rv1, 0.9 + 0.1 * var64 * n / (n - 1), 2e-6, "running var")
    # a buffer that belongs to another tensor is refused loudly (NaN statistics), never merged silently
    y2, rec2 = ops.conv2d_fwd(x[:1], w, 1, 1, stats=True)
    bad = ops.bn_stats(y, gamma, beta, stats_in=rec2)
    assert bool(torch.isnan(bad[0]).all())
    # layers without the Winograd epilogue report that they have no fused statistics
    assert ops.conv2d_fwd(x, synth_feat((Cout, Cin, 1, 1), 36).cuda(), 1, 0, stats=True)[1] is None
    with _hip.options(NO_WINO4=1):
        assert ops.conv2d_fwd(x, w, 1, 1, stats=True)[1] is None


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("cfg", [(2, 64, 18, 75, 64), (6, 128, 9, 100, 32), (5, 256, 5, 47, 64), (24, 64, 3, 94, 32),
                                 (3, 128, 4, 33, 64), (7, 64, 9, 375, 128)])
def test_batchnorm_backward_sums_from_the_dgrad_epilogue(ops, cfg, split):
    """Round 4: ``conv2d_dgrad(..., bn=...)`` + ``bn_bwd(..., sums_in=...)``: the data gradient is bit-identical to
    the plain call (the sums are a by-product), and dgamma / dbeta / dx of the BatchNorm backward agree with the
    BatchNorm's own two passes over the same tensors to 2e-6 of scale (fp32 records of <= 192 products, merged in fp64;
    both use the same ReLU mask and xhat arithmetic) and with fp64 autograd of relu(batch_norm(x)) to 2e-5.  Edge
    tiles, rows beyond H, 4-row tiles, 2 x 8 groups, cut items, and an accumulate operand (lower halves of cut items
    carry the partner's sums AND the accumulate)."""
    from asvspoof2021_air_amd import _hip
    B, Cout, H, W, Cin = cfg   # the forward conv maps Cin -> Cout; dx has Cin channels
    w = synth_feat((Cout, Cin, 3, 3), 42, scale=0.1).cuda()
    dy = synth_feat((B, Cout, H, W), 43).cuda()
    x = (synth_feat((B, Cin, H, W), 44) * 1.3 + 0.2).cuda()      # the BatchNorm's input
    acc = synth_feat((B, Cin, H, W), 45).cuda()
    gamma, beta = (1.0 + 0.3 * synth_feat((Cin,), 46)).cuda(), (0.2 * synth_feat((Cin,), 47)).cuda()
    mean, invstd, _, _ = ops.bn_stats(x, gamma, beta)
    with _hip.options(WINO4_SPLIT=split):
        for accumulate in (None, acc):
            dA0 = ops.conv2d_dgrad(dy, w, x.shape, 1, 1, accumulate=accumulate)
            dA1, sums = ops.conv2d_dgrad(dy, w, x.shape, 1, 1, accumulate=accumulate, bn=(x, mean, invstd, gamma, beta))
            assert sums is not None and torch.equal(dA0, dA1)
            dx0, dg0, db0 = ops.bn_bwd(x, dA0, mean, invstd, gamma, beta, relu=True)
            dx1, dg1, db1 = ops.bn_bwd(x, dA1, mean, invstd, gamma, beta, relu=True, sums_in=sums)
            close(dg1, dg0, 2e-6, "dgamma vs own pass")
            close(db1, db0, 2e-6, "dbeta vs own pass")
            close(dx1, dx0, 2e-6, "dx vs own pass")
            xd = x.double().requires_grad_(True)
            gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
            F.relu(F.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5)).backward(dA1.double())
            close(dg1, gd.grad, 2e-5, "dgamma vs fp64")
            close(db1, bd.grad, 2e-5, "dbeta vs fp64")
            close(dx1, xd.grad, 2e-5, "dx vs fp64")
    # sums of another tensor are refused loudly; layers without the Winograd data gradient report None
    _, s2 = ops.conv2d_dgrad(dy[:1], w, x[:1].shape, 1, 1, bn=(x[:1].contiguous(), mean, invstd, gamma, beta))
    assert bool(torch.isnan(ops.bn_bwd(x, dA0, mean, invstd, gamma, beta, relu=True, sums_in=s2)[1]).all())
    with _hip.options(NO_WINO4=1):
        assert ops.conv2d_dgrad(dy, w, x.shape, 1, 1, bn=(x, mean, invstd, gamma, beta))[1] is None
