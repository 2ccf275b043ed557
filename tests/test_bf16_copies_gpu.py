"""Round-2 entry points of the bf16 ECAPA path: operand copies written by producers, the two-operand dgrad
epilogue, the 256-wide tile kernels at even / odd T, and the BatchNorm access paths (8-byte-aligned planes, odd
planes, planes of one element).  Bit-exact where the contract is "same rounding, same arithmetic"."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.filler import synth_feat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from asvspoof2021_air_amd import ops
    return ops


def bf16_bits(t):
    """fp32 tensor -> int16 tensor of its bf16 (nearest even) bit patterns."""
    return t.detach().float().cpu().to(torch.bfloat16).view(torch.int16)


def close(got, want, rtol, name=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, "%s: rel-to-max err %.3g > %.3g" % (name, err, rtol)


@pytest.mark.parametrize("T", [750, 96, 75])
def test_cvt_and_wgrad_pre_bit_identical(ops, T):
    """air_conv1d_cvt_bf16 = bf16(x) with zero padding; a weight gradient fed with the copies (dense, and a channel
    slice of a wider copy) is bit-identical to the one that converts inside."""
    B, C = 3, 256
    x = synth_feat((B, 2 * C, T), 1).cuda()
    dy = synth_feat((B, C, T), 2).cuda()
    xb = ops.conv1d_cvt_bf16(x, ops.bf16_rows(None, B, 2 * C, T, x.device))
    Tp = xb.shape[2]
    assert Tp % 128 == 0 and Tp >= T
    assert torch.equal(xb[:, :, :T].cpu(), bf16_bits(x))
    assert int(xb[:, :, T:].abs().max()) == 0 if Tp > T else True
    dyb = ops.conv1d_cvt_bf16(dy, ops.bf16_rows(None, B, C, T, x.device))
    xs = x[:, C:]  # channel-slice view and the matching slice of the wide copy
    ref = ops.conv1d_wgrad(xs, dy, (C, C, 1), bf16=True)
    got = ops.conv1d_wgrad(xs, dy, (C, C, 1), bf16=True, x_bf=xb[:, C:], dy_bf=dyb)
    assert torch.equal(ref, got)
    got2 = ops.conv1d_wgrad(xs, dy, (C, C, 1), bf16=True, dy_bf=dyb)
    assert torch.equal(ref, got2)


@pytest.mark.parametrize("T", [750, 100])
def test_bn_bwd_writes_bf16_copy(ops, T):
    B, C = 4, 128
    x = synth_feat((B, C, T), 1).relu()
    dy = synth_feat((B, C, T), 2)
    g = 1.0 + 0.3 * synth_feat((C,), 3)
    b = 0.2 * synth_feat((C,), 4)
    xg = x.cuda()
    mean, invstd, _, _ = ops.bn_stats(xg, g.cuda(), b.cuda())
    dx0, dg0, db0 = ops.bn_bwd(xg, dy.cuda(), mean, invstd, g.cuda(), b.cuda(), relu_in=True)
    buf = ops.bf16_rows(None, B, C, T, xg.device)
    dx1, dg1, db1 = ops.bn_bwd(xg, dy.cuda(), mean, invstd, g.cuda(), b.cuda(), relu_in=True, dx_bf16=buf)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(buf[:, :, :T].cpu(), bf16_bits(dx1))
    assert int(buf[:, :, T:].abs().max()) == 0


def test_row_stats_bwd_and_asp_bwd_write_bf16_copy(ops):
    B, C, T = 2, 96, 750
    x = synth_feat((B, C, T), 1).relu().cuda()
    mean, std = ops.row_stats(x, True, 1e-4)
    dm, ds = synth_feat((B, C), 2).cuda(), synth_feat((B, C), 3).cuda()
    base = synth_feat((B, C, T), 4).cuda()
    d0 = ops.row_stats_bwd(x, mean, std, dm, ds, base.clone(), accumulate=True, relu_mask=True)
    buf = ops.bf16_rows(None, B, C, T, x.device)
    d1 = ops.row_stats_bwd(x, mean, std, dm, ds, base.clone(), accumulate=True, relu_mask=True, dx_bf16=buf)
    assert torch.equal(d0, d1) and torch.equal(buf[:, :, :T].cpu(), bf16_bits(d1))
    logits = synth_feat((B, C, T), 5).cuda()
    w0 = logits.clone()
    pooled = ops.asp_fwd(x, w0)
    dp = synth_feat((B, 2 * C), 6).cuda()
    w1 = w0.clone()
    dxa = ops.asp_bwd(x, w0, pooled, dp, torch.empty_like(x))
    buf2 = ops.bf16_rows(None, B, C, T, x.device)
    dxb = ops.asp_bwd(x, w1, pooled, dp, torch.empty_like(x), dlogits_bf16=buf2)
    assert torch.equal(dxa, dxb) and torch.equal(w0, w1)
    assert torch.equal(buf2[:, :, :T].cpu(), bf16_bits(w1))


def test_fwd_ex_writes_bf16_copy_of_the_output(ops):
    """Wide layer (GEMM epilogue writes the copy) and narrow layer (conversion pass behind the kernel)."""
    for (Cin, Cout, T) in ((1536, 1536, 750), (512, 512, 300)):
        B = 2
        x = synth_feat((B, Cin, T), 1).cuda()
        w = synth_feat((Cout, Cin, 1), 2, scale=0.05).cuda()
        bias = synth_feat((Cout,), 3, scale=0.2).cuda()
        y0 = ops.conv1d_fwd(x, w, bias, relu=True, bf16=True)
        buf = ops.bf16_rows(None, B, Cout, T, x.device)
        y1 = ops.conv1d_fwd(x, w, bias, relu=True, bf16=True, y_bf=buf)
        assert torch.equal(y0, y1)
        assert torch.equal(buf[:, :, :T].cpu(), bf16_bits(y1))
        assert int(buf[:, :, T:].abs().max()) == 0


@pytest.mark.parametrize("T", [750, 200, 75])
def test_dgrad_two_accumulate_operands(ops, T):
    """dx = dgrad + accumulate (a channel slice of a wider tensor) + accumulate2 vs the same sum formed outside;
    T = 75: odd rows, the register-staged kernel takes it."""
    B, C = 3, 512
    dy = synth_feat((B, C, T), 1).cuda()
    w = synth_feat((C, C, 1), 2, scale=0.05).cuda()
    wide = synth_feat((B, 3 * C, T), 3).cuda()
    other = synth_feat((B, C, T), 4).cuda()
    plain = ops.conv1d_dgrad(dy, w, bf16=True)
    got = ops.conv1d_dgrad(dy, w, bf16=True, accumulate=wide[:, C:2 * C], accumulate2=other)
    want = plain.double() + wide[:, C:2 * C].double() + other.double()
    close(got, want, 1e-6, "dgrad + 2 operands")
    got1 = ops.conv1d_dgrad(dy, w, bf16=True, accumulate=other)
    close(got1, plain.double() + other.double(), 1e-6, "dgrad + 1 operand")


@pytest.mark.parametrize("cfg", [(2, 512, 512, 750), (2, 512, 512, 514), (3, 512, 256, 130), (2, 256, 512, 77),
                                 (2, 1536, 128, 750), (2, 128, 1536, 258), (2, 1536, 1536, 300)])
def test_pointwise_bf16_tiles_vs_bf16_reference(ops, cfg):
    """Every tile variant (256x256, 256x128, 128x128 persistent kernels; the register-staged kernel at odd T; the
    256x256 / 128x128 GEMMs) against fp64 convolution of bf16-rounded operands: summation order only (2e-5)."""
    B, Cin, Cout, T = cfg
    x = synth_feat((B, Cin, T), 1)
    w = synth_feat((Cout, Cin, 1), 2, scale=0.05)
    b = synth_feat((Cout,), 3, scale=0.2)
    dy = synth_feat((B, Cout, T), 4)
    rnd = lambda t: t.to(torch.bfloat16).double()
    y = F.relu(F.conv1d(rnd(x), rnd(w), b.double()))
    close(ops.conv1d_fwd(x.cuda(), w.cuda(), b.cuda(), relu=True, bf16=True), y, 2e-5, "fwd")
    dx = F.conv_transpose1d(rnd(dy), rnd(w))
    close(ops.conv1d_dgrad(dy.cuda(), w.cuda(), bf16=True), dx, 2e-5, "dgrad")
    dw = torch.einsum("bot,bit->oi", rnd(dy), rnd(x)).unsqueeze(2)
    close(ops.conv1d_wgrad(x.cuda(), dy.cuda(), (Cout, Cin, 1), bf16=True), dw, 2e-5, "wgrad")


@pytest.mark.parametrize("shape", [(128, 3072, 1), (64, 128, 1), (5, 33, 7), (4, 16, 750), (3, 8, 3375), (2, 4, 751)])
def test_batchnorm_access_paths(ops, shape):
    """Planes of one element (flat kernels), short planes, 8-byte-aligned even planes, odd planes (head peeled)."""
    B, C, S = shape
    x = synth_feat(shape, 1) * 1.5 + 0.3
    gamma = 1.0 + 0.3 * synth_feat((C,), 2)
    beta = 0.2 * synth_feat((C,), 3)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.relu(F.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5))
    dy = synth_feat(shape, 6)
    y.backward(dy.double())
    xg = x.cuda()
    mean, invstd, scale, shift = ops.bn_stats(xg, gamma.cuda(), beta.cuda())
    close(ops.bn_apply(xg, scale, shift, True), y, 2e-5, "bn fwd")
    dx, dg, db = ops.bn_bwd(xg, dy.cuda(), mean, invstd, gamma.cuda(), beta.cuda(), True)
    close(dx, xd.grad, 1e-4, "bn dx")
    close(dg, gd.grad, 1e-4, "bn dgamma")
    close(db, bd.grad, 1e-4, "bn dbeta")


def test_small_linear_and_row_sums(ops):
    """linear_dx / linear_dw / sum_rows split their reduction over four waves: shapes that do not divide."""
    for (M, K, N) in ((128, 128, 512), (130, 70, 33), (7, 3072, 256)):
        x = synth_feat((M, K), 1)
        w = synth_feat((N, K), 2, scale=0.1)
        dy = synth_feat((M, N), 3)
        dx, dw, db = ops.linear_bwd(x.cuda(), w.cuda(), dy.cuda(), True)
        close(dx, dy.double() @ w.double(), 2e-5, "linear dx")
        close(dw, dy.double().t() @ x.double(), 2e-5, "linear dw")
        close(db, dy.double().sum(0), 2e-5, "linear db")
        close(ops.sum_rows(dy.cuda()), dy.double().sum(0), 2e-5, "sum_rows")


def test_tap_weights_packed_in_one_launch(ops):
    """Seven equally shaped K = 3 weights at a regular stride (as in the parameter arena), packed once: forward and
    dgrad with the packed blocks are bit-identical to the calls that pack per layer."""
    B, C, T, n = 2, 64, 300, 7
    flat = (synth_feat((n * (C * C * 3 + C),), 1, scale=0.1)).cuda()
    ws = [flat[i * (C * C * 3 + C):i * (C * C * 3 + C) + C * C * 3].view(C, C, 3) for i in range(n)]
    x = synth_feat((B, C, T), 2).cuda()
    wp = ops.conv1d_tap_pack(ws, transpose=False)
    wpt = ops.conv1d_tap_pack(ws, transpose=True)
    assert wp is not None and wp.shape == (n, 3 * C * C)
    for i in (0, 3, 6):
        y0 = ops.conv1d_fwd(x, ws[i], relu=True, dil=3, pad=3, bf16=True)
        y1 = ops.conv1d_fwd(x, ws[i], relu=True, dil=3, pad=3, bf16=True, w_packed=wp[i])
        assert torch.equal(y0, y1)
        d0 = ops.conv1d_dgrad(x, ws[i], 3, 3, bf16=True)
        d1 = ops.conv1d_dgrad(x, ws[i], 3, 3, bf16=True, w_packed=wpt[i])
        assert torch.equal(d0, d1)
    # irregular stride (parameters re-assigned outside the arena): one launch per weight, same blocks (the
    # resident path indexes the result, it must never be None)
    irr = ops.conv1d_tap_pack([ws[0], ws[2], ws[1]], transpose=False)
    assert torch.equal(irr[0], wp[0]) and torch.equal(irr[1], wp[2]) and torch.equal(irr[2], wp[1])
    assert ops.conv1d_tap_pack([ws[0].repeat(1, 1, 2)[:, :, :5].contiguous()], transpose=False) is None  # K != 3


@pytest.mark.parametrize("shape", [(4, 32, 750), (3, 16, 375), (2, 8, 1024)])
def test_bn_apply_rowmean(ops, shape):
    B, C, S = shape
    x = synth_feat(shape, 1).cuda()
    scale, shift = (1.0 + 0.3 * synth_feat((C,), 2)).cuda(), (0.2 * synth_feat((C,), 3)).cuda()
    m = torch.empty((B, C), device="cuda")
    buf = ops.bf16_rows(None, B, C, S, x.device)
    y = ops.bn_apply(x, scale, shift, relu=True, rowmean=m, y_bf=buf)
    assert torch.equal(y, ops.bn_apply(x, scale, shift, relu=True))
    close(m, y.double().mean(2), 2e-6, "row mean")
    assert torch.equal(buf[:, :, :S].cpu(), bf16_bits(y))


@pytest.mark.parametrize("T", [750, 101])
def test_res2_and_add_strided_write_bf16_slices(ops, T):
    """The Res2 chain step and the pass-through copy fill channel slices of the concat AND of its bf16 copy."""
    B, w, C = 3, 64, 192
    r = synth_feat((B, w, T), 1).cuda()
    o1 = synth_feat((B, C, T), 2).cuda()
    scale, shift = (1.0 + 0.3 * synth_feat((w,), 3)).cuda(), (0.2 * synth_feat((w,), 4)).cuda()
    cat0, cat1 = torch.zeros_like(o1), torch.zeros_like(o1)
    cat_bf = ops.bf16_rows(None, B, C, T, r.device)
    t0 = torch.empty_like(r); t1 = torch.empty_like(r)
    ops.res2_bn_apply(r, scale, shift, cat0[:, w:2 * w], o1[:, 2 * w:], t0)
    ops.res2_bn_apply(r, scale, shift, cat1[:, w:2 * w], o1[:, 2 * w:], t1, y1_bf=cat_bf[:, w:2 * w])
    assert torch.equal(cat0, cat1) and torch.equal(t0, t1)
    want = r.double() * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1)
    close(cat1[:, w:2 * w], want, 1e-6, "res2 y1")
    close(t1, want + o1[:, 2 * w:].double(), 1e-6, "res2 y2")
    ops.add_strided(cat1[:, 2 * w:], o1[:, 2 * w:], out_bf=cat_bf[:, 2 * w:])
    assert torch.equal(cat1[:, 2 * w:], o1[:, 2 * w:])
    assert torch.equal(cat_bf[:, w:, :T].cpu(), bf16_bits(cat1[:, w:]))
    assert int(cat_bf[:, :, T:].abs().max()) == 0  # (the first group was never written: fresh memory, only the padding is zeroed)


@pytest.mark.parametrize("cfg", [(2, 1536, 1536, 750), (3, 512, 512, 750), (2, 512, 256, 200), (2, 256, 1536, 130)])
def test_pointwise_from_kmajor_bf16_copy(ops, cfg):
    """Forward and data gradient straight from the operand's bf16 copy [b][C][Tp] (K-major B operand read with
    ds_read_b64_tr_b16) vs the kernels that start from the fp32 tensor: same rounding, same products."""
    B, Cin, Cout, T = cfg
    x = synth_feat((B, Cin, T), 1).cuda()
    w = synth_feat((Cout, Cin, 1), 2, scale=0.05).cuda()
    bias = synth_feat((Cout,), 3, scale=0.2).cuda()
    dy = synth_feat((B, Cout, T), 4).cuda()
    other = synth_feat((B, Cin, T), 5).cuda()
    xb = ops.conv1d_cvt_bf16(x, ops.bf16_rows(None, B, Cin, T, x.device))
    dyb = ops.conv1d_cvt_bf16(dy, ops.bf16_rows(None, B, Cout, T, x.device))
    y_ref = ops.conv1d_fwd(x, w, bias, relu=True, bf16=True)
    ybf = ops.bf16_rows(None, B, Cout, T, x.device)
    y = ops.conv1d_pointwise_kmajor(xb, w, T, bias=bias, relu=True, y_bf=ybf)
    if Cout % 256 == 0 and Cin % 64 == 0:
        assert y is not None
        close(y, y_ref, 2e-6, "kmajor fwd")
        assert torch.equal(ybf[:, :, :T].cpu(), bf16_bits(y))
    d_ref = ops.conv1d_dgrad(dy, w, bf16=True, accumulate=other)
    d = ops.conv1d_pointwise_kmajor(dyb, w, T, dgrad=True, accumulate=other)
    if Cin % 256 == 0 and Cout % 64 == 0:
        assert d is not None
        close(d, d_ref, 2e-6, "kmajor dgrad")
        # two accumulate operands, one of them a channel slice of a wider tensor (its own batch stride)
        wide_acc = torch.cat((other, other * 0.5), 1)
        d2 = ops.conv1d_pointwise_kmajor(dyb, w, T, dgrad=True, accumulate=wide_acc[:, Cin:], accumulate2=other)
        close(d2, d_ref.double() + 0.5 * other.double(), 2e-6, "kmajor dgrad + 2 operands")
    # a channel slice of a wider copy as the operand
    wide = ops.conv1d_cvt_bf16(torch.cat((other, x), 1), ops.bf16_rows(None, B, 2 * Cin, T, x.device))
    y2 = ops.conv1d_pointwise_kmajor(wide[:, Cin:], w, T, bias=bias, relu=True)
    if y is not None:
        assert torch.equal(y2, y)
