"""GPU parity of the bf16-RESIDENT ECAPA kernels (csrc/ecapa_bf16.hip, the ``air_h_*`` entry points of
conv1d_bf16.hip) through the C-ABI wrappers of asvspoof2021_air_amd/ops_h.py.

Every kernel reads bf16, computes in fp32 and stores each value rounded once to bf16.  The reference is an fp64
evaluation of the same formula ON THE bf16 INPUT VALUES; a stored value is right when it lies within half a bf16
ulp of that exact value (plus the fp32 evaluation's own noise: 2 % of an ulp) - i.e. it is one of the two bf16
neighbours a correctly rounded result can be.  fp32 outputs (statistics, parameter gradients) hold 2e-5 of scale,
the bound of every fp32 kernel test.  Padding frames T .. Tp - 1 must come out zero."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.filler import synth_feat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oh():
    from asvspoof2021_air_amd import ops_h
    return ops_h


def res(oh, x):
    """fp32 (B, C, T) CPU -> (resident GPU tensor, the bf16 values as fp64 CPU)."""
    B, C, T = x.shape
    xb = x.to(torch.bfloat16)
    r = oh.rows(B, C, T, "cuda", zero=True)
    r[:, :, :T] = xb.view(torch.int16).cuda()
    return r, xb.double()


def val(r, T):
    assert int(r[:, :, T:].abs().max()) == 0 if r.shape[2] > T else True, "padding frames must stay zero"
    return r[:, :, :T].contiguous().view(torch.bfloat16).cpu().double()


def ulp_ok(got, exact, name, slack=0.52, floor=None):
    got, exact = got.numpy(), exact.numpy()
    # bf16: 8 significant bits -> ulp(x) = 2^(floor(log2|x|) - 7).  Results far below the tensor's scale come from
    # cancelling fp32 terms, whose rounding is relative to the TERMS: the ulp is floored at 1e-5 of the scale.
    if floor is None:
        floor = max(float(np.abs(exact).max()) * 1e-5, 1e-30)
    mag = np.maximum(np.abs(exact), floor)
    ulp = 2.0 ** (np.floor(np.log2(mag)) - 7)
    err = np.abs(got - exact) / ulp
    bad = err > slack
    assert not bad.any(), "%s: %d of %d values off by more than %.2f ulp (worst %.3f ulp at exact %.6g got %.6g)" % (
        name, bad.sum(), bad.size, slack, err.max(), exact.flat[err.argmax()], got.flat[err.argmax()])


def close32(got, want, name, rtol=2e-5):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    s = max(np.abs(want).max(), 1e-30)
    assert np.abs(got - want).max() <= rtol * s, "%s: %.3g of scale %.3g" % (name, np.abs(got - want).max(), s)


SHAPES = [(3, 64, 96), (2, 128, 750), (5, 64, 401), (150, 8, 40)]  # the last: 150 splits per channel (three lane chunks)


@pytest.mark.parametrize("shape", SHAPES)
def test_bn_stats_apply_bwd(oh, shape):
    B, C, T = shape
    x, xd = res(oh, F.relu(synth_feat(shape, 1) * 1.5 + 0.3))
    gamma = 1.0 + 0.3 * synth_feat((C,), 2)
    beta = 0.2 * synth_feat((C,), 3)
    rm, rv = 0.1 * synth_feat((C,), 4), 1.0 + 0.1 * synth_feat((C,), 5).abs()
    rmg, rvg = rm.cuda(), rv.cuda()
    mean, invstd, scale, shift = oh.bn_stats(x, T, gamma.cuda(), beta.cuda(), rmg, rvg)
    xg = xd.clone().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(xg, rm64, rv64, gd, bd, True, 0.1, 1e-5)
    close32(mean, xd.mean((0, 2)), "mean")
    close32(invstd, 1.0 / torch.sqrt(xd.var((0, 2), unbiased=False) + 1e-5), "invstd")
    close32(rmg, rm64, "running_mean")
    close32(rvg, rv64, "running_var")
    rowmean = torch.empty((B, C), device="cuda")
    yh = oh.bn_apply(x, T, scale, shift, rowmean=rowmean)
    ulp_ok(val(yh, T), y.detach(), "bn_apply")
    close32(rowmean, val(yh, T).mean(2), "rowmean of the stored values")
    # backward of conv -> ReLU -> BN with the joins: dy + dy2 + rowbias / T, ReLU mask, bias gradient
    dy, dyd = res(oh, synth_feat(shape, 6))
    dy2, dy2d = res(oh, 0.5 * synth_feat(shape, 7))
    rb = synth_feat((B, C), 8)
    g_in = dyd + dy2d + rb.double()[:, :, None] / T
    y.backward(g_in)
    mask = (xd > 0).double()
    want_dx = xg.grad * mask
    dgamma, dbeta, dbias = (torch.empty(C, device="cuda") for _ in range(3))
    dx = oh.bn_bwd(x, dy, T, mean, invstd, gamma.cuda(), dgamma, dbeta, dy2=dy2, rowbias=rb.cuda(), rowbias_scale=1.0 / T,
                   dbias=dbias, relu_in=True)
    # the kernel forms xhat from the fp32 mean / invstd: allow their rounding on top of the half ulp
    ulp_ok(val(dx, T), want_dx, "bn_bwd dx", slack=0.56, floor=float(want_dx.abs().max()) * 2e-3)
    close32(dgamma, gd.grad, "dgamma", 1e-4)
    close32(dbeta, bd.grad, "dbeta", 1e-4)
    close32(dbias, want_dx.sum((0, 2)), "dbias", 1e-4)
    # in place (dx aliases dy), slices of a wider tensor
    wide = oh.rows(B, 2 * C, T, "cuda", zero=True)
    wide[:, C:] = dy
    oh.bn_bwd(x, wide[:, C:], T, mean, invstd, gamma.cuda(), dgamma, dbeta, dx=wide[:, C:], dy2=dy2, rowbias=rb.cuda(),
              rowbias_scale=1.0 / T, relu_in=True)
    assert torch.equal(wide[:, C:], dx)


@pytest.mark.parametrize("shape", SHAPES)
def test_res2_se_copy(oh, shape):
    B, C, T = shape
    x, xd = res(oh, synth_feat(shape, 11))
    add, addd = res(oh, synth_feat(shape, 12))
    scale, shift = 1.0 + 0.2 * synth_feat((C,), 13), 0.3 * synth_feat((C,), 14)
    wide = oh.rows(B, 3 * C, T, "cuda", zero=True)
    y2 = oh.rows(B, C, T, "cuda")
    oh.res2_bn_apply(x, T, scale.cuda(), shift.cuda(), wide[:, C:2 * C], add=add, y2=y2)
    want1 = xd * scale.double()[None, :, None] + shift.double()[None, :, None]
    ulp_ok(val(wide[:, C:2 * C].contiguous(), T), want1, "res2 y1")
    y1 = val(wide[:, C:2 * C].contiguous(), T)
    ulp_ok(val(y2, T), y1 + addd, "res2 y2 = bf16(stored y1 + add)")
    assert int(wide[:, :C].abs().max()) == 0 and int(wide[:, 2 * C:].abs().max()) == 0
    # SE gate + residual and its backward
    z = synth_feat((B, C), 15)
    out = oh.rows(B, C, T, "cuda")
    oh.se_scale_fwd(x, z.cuda(), add, T, out)
    g = torch.sigmoid(z.double())[:, :, None]
    ulp_ok(val(out, T), xd * g + addd, "se_scale_fwd")
    dout, doutd = res(oh, synth_feat(shape, 16))
    dx, dz = oh.se_scale_bwd(x, z.cuda(), dout, T)
    ulp_ok(val(dx, T), doutd * g, "se_scale_bwd dx")
    close32(dz, (doutd * xd).sum(2) * (g * (1 - g))[:, :, 0], "se_scale_bwd dz", 1e-4)
    # layout edges
    close32(oh.to_f32(x, T), xd, "to_f32", 0.0)
    cp = oh.rows(B, C, T, "cuda")
    oh.copy(wide[:, C:2 * C], cp)
    assert torch.equal(cp, wide[:, C:2 * C])
    back = oh.from_f32(xd.float().cuda())
    assert torch.equal(back, x)


@pytest.mark.parametrize("shape", [(2, 96, 96), (2, 64, 750)])
def test_row_stats_asp(oh, shape):
    B, C, T = shape
    x, xd = res(oh, F.relu(synth_feat(shape, 21)))
    mean, std = oh.row_stats(x, T)
    close32(mean, xd.mean(2), "row mean")
    close32(std, torch.sqrt(xd.var(2).clamp(min=1e-4)), "row std")
    dmean, dstd = synth_feat((B, C), 22), synth_feat((B, C), 23)
    dx, dxd = res(oh, synth_feat(shape, 24))
    xg = xd.clone().requires_grad_(True)
    (xg.mean(2) * dmean.double()).sum().backward()
    g1 = xg.grad.clone()
    xg.grad = None
    (torch.sqrt(xg.var(2).clamp(min=1e-4)) * dstd.double()).sum().backward()
    want = (dxd + g1 + xg.grad) * (xd > 0).double()
    rowsum = torch.empty((B, C), device="cuda")
    oh.row_stats_bwd(x, T, mean, std, dmean.cuda(), dstd.cuda(), dx, accumulate=True, relu_mask=True, rowsum=rowsum)
    ulp_ok(val(dx, T), want, "row_stats_bwd", slack=0.56, floor=float(want.abs().max()) * 2e-3)
    close32(rowsum, val(dx, T).sum(2), "rowsum of the stored values", 1e-5)
    # attentive statistics pooling
    lg, lgd = res(oh, 2.0 * synth_feat(shape, 25))
    w = lg.clone()
    out = oh.asp_fwd(x, w, T)
    wd = torch.softmax(lgd, 2)
    ulp_ok(val(w, T), wd, "asp softmax", slack=0.56)
    ws = val(w, T)  # the stored weights define the pooled statistics
    mu = (xd * ws).sum(2)
    sg = torch.sqrt(((xd ** 2) * ws).sum(2) - mu ** 2).clamp(min=1e-2)
    sg = torch.sqrt((((xd ** 2) * ws).sum(2) - mu ** 2).clamp(min=1e-4))
    close32(out[:, :C], mu, "asp mu")
    close32(out[:, C:], sg, "asp sg", 1e-4)
    # backward: stored w as a leaf
    dout = synth_feat((B, 2 * C), 26)
    xl = xd.clone().requires_grad_(True)
    al = lgd.clone().requires_grad_(True)
    wl = torch.softmax(al, 2)
    # the kernel differentiates through softmax at the STORED weights: evaluate the analytic formula there
    dm_, ds_ = dout.double()[:, :C], dout.double()[:, C:]
    dq = torch.where(sg * sg > 1e-4, ds_ / (2 * sg), torch.zeros_like(sg))
    dmm = dm_ - 2 * mu * dq
    dwv = dmm[:, :, None] * xd + dq[:, :, None] * xd ** 2
    want_dx = dmm[:, :, None] * ws + 2 * dq[:, :, None] * xd * ws
    dot = (ws * dwv).sum(2, keepdim=True)
    want_da = ws * (dwv - dot)
    dxo = oh.rows(B, C, T, "cuda")
    rs = torch.empty((B, C), device="cuda")
    oh.asp_bwd(x, w, T, out, dout.cuda(), dxo, rowsum=rs)
    ulp_ok(val(dxo, T), want_dx, "asp_bwd dx", slack=0.6, floor=float(want_dx.abs().max()) * 2e-3)
    ulp_ok(val(w, T), want_da, "asp_bwd dlogits", slack=0.6, floor=float(want_da.abs().max()) * 2e-3)
    # (analytically zero - softmax over T is shift-invariant: compare on the scale of the summed magnitudes)
    assert float((rs.cpu().double() - val(w, T).sum(2)).abs().max()) <= 1e-5 * float(val(w, T).abs().sum(2).max())


POINTWISE = [(2, 512, 512, 750), (3, 1536, 1536, 96), (2, 1536, 128, 401), (2, 128, 1536, 401), (4, 512, 512, 96)]


@pytest.mark.parametrize("cfg", POINTWISE)
def test_pointwise_conv_fwd_dgrad_wgrad(oh, cfg):
    B, Cin, Cout, T = cfg
    x, xd = res(oh, synth_feat((B, Cin, T), 31))
    w = synth_feat((Cout, Cin, 1), 32, scale=0.05)
    wb = w.to(torch.bfloat16).double()
    bias, bbc = 0.1 * synth_feat((Cout,), 33), 0.1 * synth_feat((B, Cout), 34)
    y = oh.conv_pointwise(x, w.cuda(), T, bias=bias.cuda(), bias_bc=bbc.cuda(), relu=True)
    want = F.relu(F.conv1d(xd, wb) + bias.double()[None, :, None] + bbc.double()[:, :, None])
    # K-long fp32 sums of exact products: their order noise (1e-6 of scale) on top of the half ulp
    ulp_ok(val(y, T), want, "pointwise fwd", slack=0.56, floor=float(want.abs().max()) * 4e-3)
    dy, dyd = res(oh, synth_feat((B, Cout, T), 35))
    a1, a1d = res(oh, synth_feat((B, Cin, T), 36))
    wide = oh.rows(B, 2 * Cin, T, "cuda", zero=True)
    a2v, a2d = res(oh, synth_feat((B, Cin, T), 37))
    wide[:, Cin:] = a2v
    dx = oh.conv_pointwise(dy, w.cuda(), T, dgrad=True, acc=a1, acc2=wide[:, Cin:])
    want = F.conv_transpose1d(dyd, wb) + a1d + a2d
    ulp_ok(val(dx, T), want, "pointwise dgrad + acc + acc2", slack=0.56, floor=float(want.abs().max()) * 4e-3)
    if Cin % 128 == 0 and Cout % 128 == 0:
        dw = torch.empty((Cout, Cin, 1), device="cuda")
        oh.conv_wgrad(x, dy, T, dw)
        close32(dw, torch.einsum("bot,bit->oi", dyd, xd).unsqueeze(2), "pointwise wgrad")


@pytest.mark.parametrize("cfg", [(3, 64, 96, 2), (2, 64, 750, 3), (2, 64, 401, 4)])
def test_tap_conv(oh, cfg):
    from asvspoof2021_air_amd import ops
    B, C, T, d = cfg
    wide = oh.rows(B, 4 * C, T, "cuda", zero=True)
    xv, xd = res(oh, synth_feat((B, C, T), 41))
    wide[:, C:2 * C] = xv
    w = synth_feat((C, C, 3), 42, scale=0.1)
    wb = w.to(torch.bfloat16).double()
    bias = 0.1 * synth_feat((C,), 43)
    wp = ops.conv1d_tap_pack([w.cuda()], transpose=False)
    y = oh.conv_tap(wide[:, C:2 * C], wp[0], T, d, C, C, bias=bias.cuda(), relu=True)
    want = F.relu(F.conv1d(xd, wb, bias.double(), 1, d, d))
    ulp_ok(val(y, T), want, "tap fwd", slack=0.56, floor=float(want.abs().max()) * 4e-3)
    dy, dyd = res(oh, synth_feat((B, C, T), 44))
    wpt = ops.conv1d_tap_pack([w.cuda()], transpose=True)
    oh.conv_tap(dy, wpt[0], T, d, C, C, dgrad=True, out=wide[:, 3 * C:])
    want = F.conv_transpose1d(dyd, wb, None, 1, d, 0, 1, d)
    ulp_ok(val(wide[:, 3 * C:].contiguous(), T), want, "tap dgrad", slack=0.56, floor=float(want.abs().max()) * 4e-3)
    assert int(wide[:, :C].abs().max()) == 0 and int(wide[:, 2 * C:3 * C].abs().max()) == 0


@pytest.mark.parametrize("cfg", [(3, 64, 512, 96), (2, 512, 512, 750), (2, 1536, 128, 401), (9, 128, 1536, 200)])
def test_pointwise_conv_statistics_from_the_epilogue(oh, cfg):
    """air_h_conv1d_pointwise_ex: the K = 1 conv -> ReLU -> BatchNorm1d pairs (ecapa_tdnn.py:67-69,87-89,148-150,159-161)
    take the BatchNorm's statistics from the GEMM's epilogue.  Stored tensor bit-identical to the plain call's; the four
    coefficient vectors and the running statistics equal the pass over the tensor to 2e-6, an fp64 evaluation to 3e-6."""
    B, Cin, Cout, T = cfg
    x, _ = res(oh, synth_feat((B, Cin, T), 71))
    w = synth_feat((Cout, Cin, 1), 72, scale=0.05).cuda()
    bias, bbc = (0.1 * synth_feat((Cout,), 73)).cuda(), (0.1 * synth_feat((B, Cout), 74)).cuda()
    gamma, beta = (1.0 + 0.2 * synth_feat((Cout,), 75)).cuda(), (0.3 * synth_feat((Cout,), 76)).cuda()
    y0 = oh.conv_pointwise(x, w, T, bias=bias, bias_bc=bbc, relu=True)
    y1, rec = oh.conv_pointwise(x, w, T, bias=bias, bias_bc=bbc, relu=True, stats=True)
    assert torch.equal(y0, y1)
    rm0, rv0 = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    rm1, rv1 = rm0.clone(), rv0.clone()
    st0 = oh.bn_stats(y0, T, gamma, beta, rm0, rv0)
    st1 = oh.bn_stats(y1, T, gamma, beta, rm1, rv1, stats_in=rec)
    yd = val(y0, T).double()
    want = [yd.mean((0, 2)), 1.0 / torch.sqrt(yd.var((0, 2), unbiased=False) + 1e-5)]
    for k, name in enumerate(("mean", "invstd", "scale", "shift")):
        a, b = st0[k].double().cpu(), st1[k].double().cpu()
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), name
        if k < 2:
            assert float((b - want[k]).abs().max()) <= 3e-6 * max(1.0, float(want[k].abs().max())), name
    assert float((rm0 - rm1).abs().max()) <= 2e-6 and float((rv0 - rv1).abs().max()) <= 2e-6
    with pytest.raises(Exception):
        oh.conv_pointwise(x, w.transpose(0, 1).contiguous(), T, dgrad=True, stats=True)


def test_pointwise_conv_statistics_large_mean(oh):
    """The epilogue records are UNSHIFTED fp32 {sum, sum of squares} of 64 stored values each, merged in fp64
    (h_bn_finalize_records_kernel): the variance s2 / N - mean^2 loses ~6e-8 (1 + mean^2 / var) relative.  A channel
    whose mean is 30 - 60 standard deviations (mean^2 / var ~ 1e3 - 4e3; ECAPA's post-ReLU channels sit at 0.5 - 2)
    still gets its mean to 3e-6 and its inverse deviation to 5e-4 of an fp64 evaluation of the stored values, and
    agrees with the two-pass statistics (shifted sums) to the same bound."""
    B, Cin, Cout, T = 4, 64, 512, 300
    x, _ = res(oh, synth_feat((B, Cin, T), 81))
    w = synth_feat((Cout, Cin, 1), 82, scale=0.05).cuda()
    bias = (30.0 + 0.1 * synth_feat((Cout,), 83)).cuda()
    gamma, beta = (1.0 + 0.2 * synth_feat((Cout,), 85)).cuda(), (0.3 * synth_feat((Cout,), 86)).cuda()
    y1, rec = oh.conv_pointwise(x, w, T, bias=bias, relu=True, stats=True)
    st0 = oh.bn_stats(y1, T, gamma, beta)
    st1 = oh.bn_stats(y1, T, gamma, beta, stats_in=rec)
    yd = val(y1, T).double()
    mean, var = yd.mean((0, 2)), yd.var((0, 2), unbiased=False)
    ratio = float((mean * mean / var).max())
    assert 500.0 < ratio < 2e4, ratio  # the test is about THIS regime
    want = [mean, 1.0 / torch.sqrt(var + 1e-5)]
    for k, (name, tol) in enumerate((("mean", 3e-6), ("invstd", 5e-4))):
        for st in (st0, st1):
            got = st[k].double().cpu()
            assert float(((got - want[k]).abs() / want[k].abs()).max()) <= tol, name


@pytest.mark.parametrize("cfg", [(3, 64, 96, 2), (2, 64, 750, 3), (5, 128, 401, 4), (130, 64, 200, 2)])
def test_tap_conv_statistics_from_the_epilogue(oh, cfg):
    """air_h_conv1d_tap_ex: the Res2 branch conv (ecapa_tdnn.py:46-48, conv -> ReLU -> BatchNorm1d) leaves the BatchNorm
    statistics of its STORED output in its epilogue; air_h_bn_stats_ex merges the records in fp64.  The stored tensor
    is bit-identical to the plain call's; mean / invstd / scale / shift and the running statistics equal the pass over
    the tensor to 2e-6 (both are fp64 merges of fp32 partial sums of the same bf16 values, grouped differently) and
    an fp64 evaluation of the stored values to 3e-6."""
    from asvspoof2021_air_amd import ops
    B, C, T, d = cfg
    x, _ = res(oh, synth_feat((B, C, T), 51))
    w = synth_feat((C, C, 3), 52, scale=0.1)
    bias = 0.1 * synth_feat((C,), 53)
    wp = ops.conv1d_tap_pack([w.cuda()], transpose=False)
    gamma, beta = (1.0 + 0.2 * synth_feat((C,), 54)).cuda(), (0.3 * synth_feat((C,), 55)).cuda()
    y0 = oh.conv_tap(x, wp[0], T, d, C, C, bias=bias.cuda(), relu=True)
    y1, rec = oh.conv_tap(x, wp[0], T, d, C, C, bias=bias.cuda(), relu=True, stats=True)
    assert torch.equal(y0, y1)
    rm0, rv0 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm1, rv1 = rm0.clone(), rv0.clone()
    st0 = oh.bn_stats(y0, T, gamma, beta, rm0, rv0)
    st1 = oh.bn_stats(y1, T, gamma, beta, rm1, rv1, stats_in=rec)
    yd = val(y0, T).double()
    mean = yd.mean((0, 2))
    var = yd.var((0, 2), unbiased=False)
    want = [mean, 1.0 / torch.sqrt(var + 1e-5)]
    for k, name in enumerate(("mean", "invstd", "scale", "shift")):
        a, b = st0[k].double().cpu(), st1[k].double().cpu()
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), name
        if k < 2:
            assert float((b - want[k]).abs().max()) <= 3e-6 * max(1.0, float(want[k].abs().max())), name
    assert float((rm0 - rm1).abs().max()) <= 2e-6 and float((rv0 - rv1).abs().max()) <= 2e-6
    with pytest.raises(Exception):  # records of another geometry are refused (half of them would be a GEMM epilogue's count)
        oh.bn_stats(y1, T, gamma, beta, stats_in=rec[:rec.numel() // 4].clone())


@pytest.mark.parametrize("cfg", [(3, 64, 96, 2), (2, 64, 750, 3), (5, 128, 401, 4), (130, 64, 200, 2)])
def test_tap_dgrad_batchnorm_backward_sums_from_the_epilogue(oh, cfg):
    """air_h_conv1d_tap_ex2 on a data-gradient launch of the Res2 chain (ecapa_tdnn.py:79-85 backward): the gradient it
    writes joins the previous branch's slice of the concat gradient on the way into that branch's BatchNorm, and the
    epilogue leaves the five sums of that BatchNorm's backward.  The stored gradient is bit-identical to the plain
    call's; dgamma / dbeta / dbias of air_h_bn_bwd_ex(sums_in) equal the three-pass form's to 3e-6 of their scale
    (fp64 merges of fp32 partial sums of the same values, grouped differently) and the dx it writes to one bf16 ulp."""
    from asvspoof2021_air_amd import ops
    B, C, T, d = cfg
    wide = oh.rows(B, 2 * C, T, "cuda", zero=True)  # [slice of branch i - 1 | slice of branch i] of the concat gradient
    dcat_prev, _ = res(oh, synth_feat((B, C, T), 61))
    wide[:, :C] = dcat_prev
    dc, _ = res(oh, synth_feat((B, C, T), 62))             # d(conv output of branch i)
    r_prev, _ = res(oh, synth_feat((B, C, T), 63).relu())  # ReLU output of branch i - 1 = its BatchNorm's input
    w = synth_feat((C, C, 3), 64, scale=0.1)
    wpt = ops.conv1d_tap_pack([w.cuda()], transpose=True)
    gamma = (1.0 + 0.2 * synth_feat((C,), 65)).cuda()
    mean, invstd, _, _ = oh.bn_stats(r_prev, T, gamma, torch.zeros(C, device="cuda"))
    plain = oh.conv_tap(dc, wpt[0], T, d, C, C, dgrad=True)
    din, sums = oh.conv_tap(dc, wpt[0], T, d, C, C, dgrad=True, out=wide[:, C:], bn=(r_prev, wide[:, :C], mean, invstd))
    assert torch.equal(plain, wide[:, C:].contiguous())
    outs = []
    for s_in in (None, sums):
        dg, db_, dbias = (torch.zeros(C, device="cuda") for _ in range(3))
        dx = oh.bn_bwd(r_prev, wide[:, :C], T, mean, invstd, gamma, dg, db_, dy2=wide[:, C:], dbias=dbias, sums_in=s_in)
        outs.append((dg, db_, dbias, dx))
    for k, name in enumerate(("dgamma", "dbeta", "dbias")):
        a, b_ = outs[0][k].double().cpu(), outs[1][k].double().cpu()
        assert float((a - b_).abs().max()) <= 3e-6 * max(1.0, float(a.abs().max())), name
    a, b_ = val(outs[0][3], T), val(outs[1][3], T)
    assert float(((a - b_).abs() / (a.abs() + 1e-3)).max()) <= 2.0 ** -7, "dx"
    with pytest.raises(Exception):
        oh.bn_bwd(r_prev, wide[:, :C], T, mean, invstd, gamma, dg, db_, dy2=wide[:, C:], sums_in=sums[:sums.numel() // 2].clone())


@pytest.mark.parametrize("cfg", [(3, 64, 96, 2, 3), (2, 64, 750, 3, 7), (5, 128, 401, 4, 2), (130, 64, 200, 2, 7)])
def test_tap_conv_wgrad_all_branches(oh, cfg):
    """air_h_conv1d_tap_wgrad: every branch of a block in one launch, operands as channel slices of wider tensors
    (batch strides), against the fp64 contraction of the same bf16 values."""
    B, W, T, d, nb = cfg
    widex = oh.rows(B, nb * W, T, "cuda", zero=True)
    xs, dys, want = [], [], []
    for i in range(nb):
        xv, xd = res(oh, synth_feat((B, W, T), 51 + i))
        widex[:, i * W:(i + 1) * W] = xv
        dy, dyd = res(oh, synth_feat((B, W, T), 71 + i))
        xs.append(widex[:, i * W:(i + 1) * W])
        dys.append(dy)
        xp = F.pad(xd, (d, d))
        want.append(torch.stack([torch.einsum("bot,bit->oi", dyd, xp[:, :, k * d:k * d + T]) for k in range(3)], dim=2))
    outs = [torch.full((W, W, 3), float("nan"), device="cuda") for _ in range(nb)]
    oh.conv_tap_wgrad(xs, dys, T, d, outs)
    for i in range(nb):
        close32(outs[i], want[i], "tap wgrad branch %d" % i)
    # run-to-run identical (fixed-order split over the utterances)
    again = [torch.empty((W, W, 3), device="cuda") for _ in range(nb)]
    oh.conv_tap_wgrad(xs, dys, T, d, again)
    assert all(torch.equal(a, b) for a, b in zip(outs, again))


# ---------------------------------------------------------------------------------------------------------------
# the TRAIN-mode analogue of the eval-mode pin.  Model-level bf16 parity in train mode has been a
# statistical band (batch statistics amplify one rounding ~100x through the filler-initialised net).  Teacher forcing
# removes the amplification without leaving train mode: ONE whole train-mode forward of the bf16-resident model keeps
# every tensor it stores (they are what backward reads); each stored tensor is then re-derived in fp64 from the stored
# tensors it was computed FROM - the model's own inputs to that layer, bit for bit - with the layer's BatchNorm
# statistics re-derived in fp64 from the stored BatchNorm input.  Required: every stored value within half a bf16 ulp
# of that evaluation (the kernel tests' slack: + the fp32 sums' order noise), every batch statistic the kernels took
# (from the convolution epilogues) within fp32 rounding of the fp64 statistic of the stored tensor.
# A BatchNorm output far below the tensor's scale is the difference of O(scale) fp32 terms ((r - mean) * invstd * gamma
# + beta with the kernel's fp32 statistics, which sit within 2e-6 of the fp64 ones): its rounding is relative to the
# TERMS.  Values below 2e-4 of the scale are held to the ulp of 2e-4 x scale (an absolute 4e-7 x scale).
BN_FLOOR = 2e-4


def _stats64(r):
    mean = r.mean((0, 2))
    var = r.var((0, 2), unbiased=False)
    return mean, 1.0 / torch.sqrt(var + 1e-5)


def _bn64(r, bn, st, name):
    """fp64 BatchNorm(train) of the stored input r; checks the kernel's (mean, invstd) against the fp64 statistics."""
    mean, invstd = _stats64(r)
    rms = float(torch.sqrt((r * r).mean()))
    got_m, got_i = st[0].cpu().double(), st[1].cpu().double()
    assert float((got_m - mean).abs().max()) <= 2e-6 * rms, "%s: batch mean %.3g of rms" % (name, float((got_m - mean).abs().max()) / rms)
    assert float((got_i / invstd - 1).abs().max()) <= 1e-5, "%s: invstd rel %.3g" % (name, float((got_i / invstd - 1).abs().max()))
    g, b = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
    return (r - mean[None, :, None]) * (invstd * g)[None, :, None] + b[None, :, None]


@pytest.mark.parametrize("B,T", [(8, 200), (3, 401)])
def test_train_mode_every_stored_tensor_teacher_forced(oh, B, T):
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from oracle.filler import fill_module_
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    fill_module_(m)
    m = m.cuda().train().set_compute_dtype("bf16")
    x = synth_feat((B, 60, T), seed=900 + T)
    with torch.no_grad():
        feat, out, S = m._forward_impl(x.cuda(), save=True)
    torch.cuda.synchronize()
    assert S["resident"] and S["T"] == T
    d = lambda p: p.detach().cpu().double()
    rb = lambda t: t.to(torch.bfloat16).double()  # bf16 rounding of operands (weights)
    V = lambda r: val(r.contiguous(), T)
    checked = [0]

    def stored(got_rows, exact, name, slack=0.56, floor_frac=4e-3):
        ulp_ok(V(got_rows), exact, name, slack=slack, floor=float(exact.abs().max()) * floor_frac)
        checked[0] += 1

    def pw(xs, conv_w, bias, relu=True, bias_bc=None):
        y = F.conv1d(xs, rb(d(conv_w)).view(conv_w.shape[0], -1, 1)) + d(bias)[None, :, None]
        if bias_bc is not None:
            y = y + bias_bc[:, :, None]
        return F.relu(y) if relu else y

    # conv1 (K = 5): bf16 operands, fp32 accumulate + fp32 bias -> ReLU -> stored; BatchNorm -> stored
    c1 = F.conv1d(rb(x.double()), rb(d(m.conv1.weight)), None, 1, 2) + d(m.conv1.bias)[None, :, None]
    stored(S["r0"], F.relu(c1), "conv1 -> relu")
    stored(S["h"], _bn64(V(S["r0"]), m.bn1, S["st0"], "bn1"), "bn1", slack=0.53, floor_frac=BN_FLOOR)
    inp = S["h"]
    for k, (blk, SB) in enumerate(zip((m.layer1, m.layer2, m.layer3), S["blocks"])):
        nm = "layer%d." % (k + 1)
        w, dil, nums = blk.width, blk.dilation, blk.nums
        assert SB["inp"].data_ptr() == inp.data_ptr()
        xin = V(SB["inp"])
        stored(SB["r1"], pw(xin, blk.conv1.weight, blk.conv1.bias), nm + "conv1 -> relu")
        o1 = _bn64(V(SB["r1"]), blk.bn1, SB["st1"], nm + "bn1")  # (slices 0 .. nums-1 are overwritten in S['cat'] later)
        cat = V(SB["cat"])
        stored(SB["t"][0], o1[:, :w], nm + "bn1 slice 0 (branch 0 input)", slack=0.53, floor_frac=BN_FLOOR)
        ulp_ok(cat[:, nums * w:], o1[:, nums * w:], nm + "bn1 pass-through slice", slack=0.53, floor=float(o1.abs().max()) * BN_FLOOR)
        for i in range(nums):
            ti = V(SB["t"][i])
            wi = rb(d(blk.convs[i].weight))
            ri = F.relu(F.conv1d(ti, wi, None, 1, dil, dil) + d(blk.convs[i].bias)[None, :, None])
            stored(SB["r"][i], ri, nm + "convs.%d -> relu" % i)
            yi = _bn64(V(SB["r"][i]), blk.bns[i], SB["st"][i], nm + "bns.%d" % i)
            ulp_ok(cat[:, i * w:(i + 1) * w], yi, nm + "bns.%d (concat slice)" % i, slack=0.53, floor=float(yi.abs().max()) * BN_FLOOR)
            checked[0] += 1
            if i + 1 < nums:
                # next branch input = bf16(stored y_i + stored o1 slice): o1's slice is not kept, so the exact slice
                # stands in for it - half an ulp OF THE SLICE VALUE on top of the half ulp of the stored sum
                o1n = o1[:, (i + 1) * w:(i + 2) * w]
                tn = cat[:, i * w:(i + 1) * w] + o1n
                ulp = lambda v: 2.0 ** (torch.floor(torch.log2(v.abs().clamp(min=1e-30))) - 7)
                err = (V(SB["t"][i + 1]) - tn).abs()
                bound = 0.53 * ulp(tn.abs().clamp(min=float(tn.abs().max()) * 1e-5)) + 0.51 * ulp(o1n)
                assert bool((err <= bound).all()), "%sbranch %d input: worst %.3f of its bound" % (
                    nm, i + 1, float((err / bound).max()))
                checked[0] += 1
        stored(SB["r3"], pw(cat, blk.conv3.weight, blk.conv3.bias), nm + "conv3 -> relu")
        o3x = _bn64(V(SB["r3"]), blk.bn3, SB["st3"], nm + "bn3")
        stored(SB["o3"], o3x, nm + "bn3", slack=0.53, floor_frac=BN_FLOOR)
        o3 = V(SB["o3"])
        close32(SB["m"], o3.mean(2), nm + "SE squeeze")
        se = blk.se.se
        z1 = F.relu(F.linear(SB["m"].cpu().double(), d(se[1].weight).view(se[1].out_channels, -1), d(se[1].bias)))
        close32(SB["z1"], z1, nm + "se.1")
        z1n = _bn64(SB["z1"].cpu().double().unsqueeze(2), se[3], SB["stS"], nm + "se.3").squeeze(2)
        close32(SB["z1n"], z1n, nm + "se.3 apply", rtol=1e-4)
        z2 = F.linear(SB["z1n"].cpu().double(), d(se[4].weight).view(se[4].out_channels, -1), d(se[4].bias))
        close32(SB["z2"], z2, nm + "se.4")
        outk = S["cat123"][:, k * 512:(k + 1) * 512]
        stored(outk, o3 * torch.sigmoid(SB["z2"].cpu().double())[:, :, None] + xin, nm + "gate * o3 + x", slack=0.53, floor_frac=BN_FLOOR)
        inp = outk
    cat123 = V(S["cat123"])
    stored(S["x4"], pw(cat123, m.layer4.weight, m.layer4.bias), "layer4 -> relu")
    x4 = V(S["x4"])
    close32(S["mean"], x4.mean(2), "context mean")
    close32(S["std"], torch.sqrt(x4.var(2).clamp(min=1e-4)), "context std", rtol=1e-4)
    w0 = d(m.attention[0].weight).view(128, -1)
    ctxb = F.linear(torch.cat((S["mean"], S["std"]), 1).cpu().double(), w0[:, 1536:])
    a1 = F.relu(F.conv1d(x4, rb(w0[:, :1536]).unsqueeze(2)) + ctxb[:, :, None] + d(m.attention[0].bias)[None, :, None])
    stored(S["a1"], a1, "attention.0 -> relu")
    stored(S["a1n"], _bn64(V(S["a1"]), m.attention[2], S["stA"], "attention.2"), "attention.2", slack=0.53, floor_frac=BN_FLOOR)
    logits = pw(V(S["a1n"]), m.attention[3].weight, m.attention[3].bias, relu=False)
    # the kernel stores the logits as bf16 and asp_fwd overwrites them with bf16(softmax over T of the STORED logits)
    wts64 = torch.softmax(rb(logits), dim=2)
    got_w = V(S["wts"])
    # a logit within rounding of a bf16 tie moves its weight by a whole ulp of the LOGIT (e^(ulp) - 1 = 0.8 % of the
    # weight = one weight ulp): weights within 1.6 ulp, and their sum over T within 1 % of 1
    ulp_ok(got_w, wts64, "asp softmax weights", slack=1.6, floor=float(wts64.max()) * 1e-3)
    assert float((got_w.sum(2) - 1).abs().max()) <= 1e-2
    mu = (x4 * got_w).sum(2)
    sg = torch.sqrt((((x4 * x4) * got_w).sum(2) - mu * mu).clamp(min=1e-4))
    close32(S["pooled"], torch.cat((mu, sg), 1), "attentive statistics (mu | sg)", rtol=1e-4)
    p5 = _bn64(S["pooled"].cpu().double().unsqueeze(2), m.bn5, S["st5"], "bn5").squeeze(2)
    close32(S["p5"], p5, "bn5", rtol=1e-4)
    close32(feat, F.linear(S["p5"].cpu().double(), d(m.fc6.weight), d(m.fc6.bias)), "fc6", rtol=1e-4)
    print("teacher-forced train-mode pin: %d stored tensors within half a bf16 ulp, 34 BatchNorm statistics within fp32 rounding" % checked[0])
    assert checked[0] == 3 * (2 + 2 * 7 + 6 + 3) + 5  # 80 stored (B, C, T) tensors


@pytest.mark.parametrize("cfg", [(3, 96, 2), (2, 750, 3), (5, 401, 4), (130, 200, 2), (2, 128, 4), (1, 5, 3)])
def test_tap_prologue_forward_is_the_unfused_sequence_bit_for_bit(oh, cfg):
    """Round 6 (air_h_conv1d_tap_pro, kind 1): the Res2 join in front of branch i - y1 = bf16(bn(r_{i-1})) into the concat
    slice, t_i = bf16(y1 + o1's slice) - computed while the conv stages its operand.  Against air_h_res2_bn_apply followed
    by air_h_conv1d_tap_ex: the conv output, its statistics records, y1 and t_i are the SAME BITS (same arithmetic, same
    rounding points; only the launch count differs), with every tensor a channel slice of a wider one, and neighbouring
    slices untouched.  Aliasing a side output with a tensor the launch reads with a halo is refused."""
    from asvspoof2021_air_amd import _hip, ops
    B, T, d = cfg
    C = 64
    r_prev, _ = res(oh, synth_feat((B, C, T), 71).relu())
    cat0 = oh.rows(B, 4 * C, T, "cuda", zero=True)  # [guard | slice i - 1 | slice i | guard]
    add, _ = res(oh, synth_feat((B, C, T), 72))
    cat0[:, 2 * C:3 * C] = add
    scale = (1.0 + 0.2 * synth_feat((C,), 73)).cuda()
    shift = (0.3 * synth_feat((C,), 74)).cuda()
    w = synth_feat((C, C, 3), 75, scale=0.1)
    bias = (0.1 * synth_feat((C,), 76)).cuda()
    wp = ops.conv1d_tap_pack([w.cuda()], transpose=False)
    # unfused
    cat_a = cat0.clone()
    t_a = oh.rows(B, C, T, "cuda")
    oh.res2_bn_apply(r_prev, T, scale, shift, cat_a[:, C:2 * C], add=cat_a[:, 2 * C:3 * C], y2=t_a)
    y_a, rec_a = oh.conv_tap(t_a, wp[0], T, d, C, C, bias=bias, relu=True, stats=True)
    # fused
    cat_b = cat0.clone()
    t_b = oh.rows(B, C, T, "cuda")
    t_b.fill_(0x7fc0)  # every element must be written
    pro = oh.res2_prologue(scale, shift, add=cat_b[:, 2 * C:3 * C], y1=cat_b[:, C:2 * C], t_out=t_b)
    y_b, rec_b = oh.conv_tap(r_prev, wp[0], T, d, C, C, bias=bias, relu=True, stats=True, pro=pro)
    assert torch.equal(t_a, t_b), "t_i"
    assert torch.equal(cat_a, cat_b), "y1 / the concat buffer"
    assert torch.equal(y_a, y_b), "conv output"
    assert torch.equal(rec_a, rec_b), "statistics records"
    assert int(cat_b[:, :C].abs().max()) == 0 and int(cat_b[:, 3 * C:].abs().max()) == 0
    # without statistics records (eval mode)
    y_c = oh.conv_tap(r_prev, wp[0], T, d, C, C, bias=bias, relu=True, pro=oh.res2_prologue(
        scale, shift, add=cat_b[:, 2 * C:3 * C], y1=cat_b[:, C:2 * C], t_out=t_b))
    assert torch.equal(y_a, y_c)
    with pytest.raises(_hip.AirError):  # y1 may not be the tensor read as `add` (halo race)
        oh.conv_tap(r_prev, wp[0], T, d, C, C, pro=oh.res2_prologue(scale, shift, add=cat_b[:, 2 * C:3 * C],
                                                                   y1=cat_b[:, 2 * C:3 * C], t_out=t_b))
    assert not oh.tap_pro_ok(128, 128) and oh.tap_pro_ok(64, 64)


@pytest.mark.parametrize("cfg", [(3, 96, 2, True), (2, 750, 3, True), (5, 401, 4, False), (130, 200, 2, True), (1, 5, 3, False)])
def test_tap_prologue_backward_is_the_unfused_sequence_bit_for_bit(oh, cfg):
    """air_h_conv1d_tap_pro, kind 2: the BatchNorm-backward apply of branch i (h_bn_bwd_apply_kernel) as the prologue of
    that branch's data-gradient conv.  Against air_h_bn_bwd_ex followed by air_h_conv1d_tap_ex2: dc_i (side output), the
    data gradient, the previous BatchNorm's backward sums, dgamma / dbeta / dbias - the same bits; with and without the
    second gradient half (the last branch has none)."""
    from asvspoof2021_air_amd import ops
    B, T, d, two = cfg
    C = 64
    r_i, _ = res(oh, synth_feat((B, C, T), 81).relu())
    r_p, _ = res(oh, synth_feat((B, C, T), 82).relu())
    dcat = oh.rows(B, 2 * C, T, "cuda", zero=True)  # [slice i - 1 | slice i] of the concat gradient
    dcat[:, :C] = res(oh, synth_feat((B, C, T), 83))[0]
    dcat[:, C:] = res(oh, synth_feat((B, C, T), 84))[0]
    dy2 = res(oh, synth_feat((B, C, T), 85))[0] if two else None
    gamma = (1.0 + 0.2 * synth_feat((C,), 86)).cuda()
    zeros = torch.zeros(C, device="cuda")
    mean_i, invstd_i, _, _ = oh.bn_stats(r_i, T, gamma, zeros)
    mean_p, invstd_p, _, _ = oh.bn_stats(r_p, T, gamma, zeros)
    w = synth_feat((C, C, 3), 87, scale=0.1)
    wpt = ops.conv1d_tap_pack([w.cuda()], transpose=True)
    # unfused
    dg_a, db_a, dbias_a = (torch.zeros(C, device="cuda") for _ in range(3))
    dc_a = oh.bn_bwd(r_i, dcat[:, C:], T, mean_i, invstd_i, gamma, dg_a, db_a, dy2=dy2, dbias=dbias_a)
    din_a, sums_a = oh.conv_tap(dc_a, wpt[0], T, d, C, C, dgrad=True, bn=(r_p, dcat[:, :C], mean_p, invstd_p))
    # fused
    dg_b, db_b, dbias_b = (torch.zeros(C, device="cuda") for _ in range(3))
    oh.bn_bwd(r_i, dcat[:, C:], T, mean_i, invstd_i, gamma, dg_b, db_b, dy2=dy2, dbias=dbias_b, apply=False)
    dc_b = oh.rows(B, C, T, "cuda")
    dc_b.fill_(0x7fc0)
    do1 = oh.rows(B, 2 * C, T, "cuda", zero=True)
    pro = oh.bn_bwd_prologue(dcat[:, C:], dy2, mean_i, invstd_i, gamma, dg_b, db_b, dc_b)
    din_b, sums_b = oh.conv_tap(r_i, wpt[0], T, d, C, C, dgrad=True, out=do1[:, C:], pro=pro,
                                bn=(r_p, dcat[:, :C], mean_p, invstd_p))
    assert torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b) and torch.equal(dbias_a, dbias_b)
    assert torch.equal(dc_a, dc_b), "dc_i"
    assert torch.equal(din_a, do1[:, C:].contiguous()), "data gradient"
    assert torch.equal(sums_a, sums_b), "BatchNorm-backward sums of the previous branch"
    assert int(do1[:, :C].abs().max()) == 0
    # the first branch of the chain has no BatchNorm in front of it: no sums
    din_c = oh.conv_tap(r_i, wpt[0], T, d, C, C, dgrad=True, pro=oh.bn_bwd_prologue(
        dcat[:, C:], dy2, mean_i, invstd_i, gamma, dg_b, db_b, dc_b))
    assert torch.equal(din_a, din_c)


@pytest.mark.parametrize("B,T", [(4, 96), (3, 401)])
def test_fused_res2_chain_train_step_equals_unfused(B, T):
    """The whole bf16-resident model, one train-mode forward + backward with the Res2 prologues on (default) and off
    (AIR_TAP_PROLOGUE=0's switch): outputs, every parameter gradient and the BatchNorm buffers are bit-identical."""
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from oracle.filler import fill_module_
    x = synth_feat((B, 60, T), seed=500 + T).cuda()
    out = {}
    for fused in (True, False, 1, 2):
        m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        fill_module_(m)
        m = m.cuda().train().set_compute_dtype("bf16")
        m.fuse_tap_prologue = {True: 3, False: 0}.get(fused, fused) if isinstance(fused, bool) else fused
        feat, o = m(x)
        (feat.square().mean() + o.square().mean()).backward()
        out[fused] = (feat.detach().clone(), o.detach().clone(),
                      {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                      {k: v.detach().clone() for k, v in m.named_buffers()})
    for other in (True, 1, 2):
        assert torch.equal(out[other][0], out[False][0]) and torch.equal(out[other][1], out[False][1])
        assert out[other][2].keys() == out[False][2].keys()
        for k in out[other][2]:
            assert torch.equal(out[other][2][k], out[False][2][k]), (other, k)
        for k in out[other][3]:
            assert torch.equal(out[other][3][k], out[False][3][k]), (other, k)


@pytest.mark.parametrize("cfg", [(3, 96, 2), (2, 750, 3), (5, 401, 4), (1, 5, 3)])
def test_tap_row_piece_staging_equals_channel_staging(oh, cfg):
    """Option TAP_ROWS: the 64 -> 64 convs stage 16-byte row pieces (transposed into LDS by 2-byte writes) instead of
    2-byte loads down the channels - forward with statistics and data gradient with the BatchNorm sums are the same bits."""
    from asvspoof2021_air_amd import _hip, ops
    B, T, d = cfg
    C = 64
    wide = oh.rows(B, 3 * C, T, "cuda", zero=True)
    wide[:, C:2 * C] = res(oh, synth_feat((B, C, T), 91))[0]
    w = synth_feat((C, C, 3), 92, scale=0.1)
    bias = (0.1 * synth_feat((C,), 93)).cuda()
    wp = ops.conv1d_tap_pack([w.cuda()], transpose=False)
    wpt = ops.conv1d_tap_pack([w.cuda()], transpose=True)
    r_p, _ = res(oh, synth_feat((B, C, T), 94).relu())
    dyp, _ = res(oh, synth_feat((B, C, T), 95))
    mean_p, invstd_p, _, _ = oh.bn_stats(r_p, T, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"))
    got = {}
    for rows_on in (0, 1):
        with _hip.options(TAP_ROWS=rows_on):
            y, rec = oh.conv_tap(wide[:, C:2 * C], wp[0], T, d, C, C, bias=bias, relu=True, stats=True)
            din, sums = oh.conv_tap(wide[:, C:2 * C], wpt[0], T, d, C, C, dgrad=True, bn=(r_p, dyp, mean_p, invstd_p))
        got[rows_on] = (y, rec, din, sums)
    for a, b_, name in zip(got[0], got[1], ("forward", "records", "dgrad", "sums")):
        assert torch.equal(a, b_), name
