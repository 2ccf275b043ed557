"""GPU parity of the conv1d entry points and the small ECAPA kernels against PyTorch-CPU fp64."""
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


def close(got, want, rtol=2e-5, name=""):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = max(np.abs(want).max(), 1e-30)
    err = np.abs(got - want).max() / scale
    assert err <= rtol, "%s: rel-to-max err %.3g > %.3g" % (name, err, rtol)


C1D = [  # (B, Cin, T, Cout, K, dil, pad)
    (2, 60, 96, 512, 5, 1, 2),     # conv1 (ragged Cin)
    (3, 512, 75, 512, 1, 1, 0),    # Bottle2neck conv1 / conv3
    (2, 64, 100, 64, 3, 2, 2),     # Res2 dilated convs
    (2, 64, 100, 64, 3, 3, 3),
    (2, 64, 33, 64, 3, 4, 4),
    (2, 1536, 40, 1536, 1, 1, 0),  # layer4
    (2, 1536, 40, 128, 1, 1, 0),   # attention.0 (x part)
    (2, 128, 40, 1536, 1, 1, 0),   # attention.3
]


@pytest.mark.parametrize("cfg", C1D)
def test_conv1d(ops, cfg):
    B, Cin, T, Cout, K, dil, pad = cfg
    x = synth_feat((B, Cin, T), 1).double().requires_grad_(True)
    w = synth_feat((Cout, Cin, K), 2, scale=0.05).double().requires_grad_(True)
    b = synth_feat((Cout,), 3, scale=0.2).double().requires_grad_(True)
    bbc = synth_feat((B, Cout), 4, scale=0.2)
    y = F.relu(F.conv1d(x, w, b, 1, pad, dil) + bbc.double().unsqueeze(2))
    dy = synth_feat(tuple(y.shape), 5)
    y.backward(dy.double())
    xg, wg = x.detach().float().cuda(), w.detach().float().cuda()
    got = ops.conv1d_fwd(xg, wg, b.detach().float().cuda(), bbc.cuda(), relu=True, dil=dil, pad=pad)
    close(got, y, name="conv1d fwd")
    dc = (dy.double() * (y > 0)).float()  # gradient at the pre-ReLU conv output
    close(ops.conv1d_wgrad(xg, dc.cuda(), (Cout, Cin, K), dil, pad), w.grad, name="conv1d wgrad")
    close(ops.channel_sum(dc.cuda()), b.grad, name="bias grad")
    if Cin % 64 == 0:
        close(ops.conv1d_dgrad(dc.cuda(), wg, dil, pad), x.grad, name="conv1d dgrad")


def test_conv1d_channel_slice_views(ops):
    """Res2 split / concat handled as views: slice in, slice out, accumulate."""
    B, T = 2, 50
    big = synth_feat((B, 512, T), 1).cuda()
    w = synth_feat((64, 64, 3), 2, scale=0.1)
    out_big = torch.zeros(B, 512, T, device="cuda")
    ops.conv1d_fwd(big[:, 128:192], w.cuda(), dil=2, pad=2, out=out_big[:, 64:128])
    want = F.conv1d(big[:, 128:192].cpu().double(), w.double(), None, 1, 2, 2)
    close(out_big[:, 64:128], want, name="slice conv")
    assert float(out_big[:, :64].abs().max()) == 0 and float(out_big[:, 128:].abs().max()) == 0
    s = torch.empty(B, 64, T, device="cuda")
    ops.add_strided(s, big[:, 0:64], big[:, 64:128])
    close(s, (big[:, 0:64] + big[:, 64:128]).cpu(), name="add_strided")
    ops.add_strided(out_big[:, 448:512], big[:, 448:512])
    assert torch.equal(out_big[:, 448:512], big[:, 448:512])


def test_row_stats_and_bwd(ops):
    B, C, T = 3, 40, 77
    x = (synth_feat((B, C, T), 1) * 0.5).double().requires_grad_(True)
    mean = x.mean(2)
    std = torch.sqrt(x.var(2).clamp(min=1e-4))
    dm, ds = synth_feat((B, C), 2).double(), synth_feat((B, C), 3).double()
    (mean * dm + std * ds).sum().backward()
    xg = x.detach().float().cuda()
    m, s = ops.row_stats(xg)
    close(m, mean, name="row mean")
    close(s, std, name="row std")
    dx = torch.zeros(B, C, T, device="cuda")
    ops.row_stats_bwd(xg, m, s, dm.float().cuda(), ds.float().cuda(), dx, accumulate=True)
    close(dx, x.grad, rtol=1e-4, name="row stats bwd")
    close(ops.row_sum(xg), x.sum(2), name="row sum")
    # fused form used after layer4 (ecapa_tdnn.py:173-178): x = relu(c); gradient w.r.t. c, plus per-row sums
    c = synth_feat((B, C, T), 4).double().requires_grad_(True)
    r = torch.relu(c)
    up = synth_feat((B, C, T), 5)
    ((r.mean(2) * dm + torch.sqrt(r.var(2).clamp(min=1e-4)) * ds).sum() + (r * up.double()).sum()).backward()
    rg = r.detach().float().cuda()
    m2, s2 = ops.row_stats(rg)
    dxf = up.cuda().clone()
    rows = torch.empty(B, C, device="cuda")
    ops.row_stats_bwd(rg, m2, s2, dm.float().cuda(), ds.float().cuda(), dxf, accumulate=True, relu_mask=True, rowsum=rows)
    close(dxf, c.grad, rtol=1e-4, name="row stats bwd + relu mask")
    close(rows, c.grad.sum(2), rtol=1e-4, name="row sums")
    close(ops.sum_rows(rows), c.grad.sum(dim=(0, 2)), rtol=1e-4, name="bias gradient")


def test_se_scale(ops):
    B, C, T = 2, 48, 61
    x = synth_feat((B, C, T), 1).double().requires_grad_(True)
    z = synth_feat((B, C), 2).double().requires_grad_(True)
    res_big = synth_feat((B, 3 * C, T), 3)
    res = res_big[:, C:2 * C].double()
    out = x * torch.sigmoid(z).unsqueeze(2) + res
    dout = synth_feat((B, C, T), 4)
    out.backward(dout.double())
    og = torch.zeros(B, 2 * C, T, device="cuda")
    ops.se_scale_fwd(x.detach().float().cuda(), z.detach().float().cuda(), res_big.cuda()[:, C:2 * C], og[:, :C])
    close(og[:, :C], out, name="se fwd")
    dx, dz = ops.se_scale_bwd(x.detach().float().cuda(), z.detach().float().cuda(), dout.cuda())
    close(dx, x.grad, name="se dx")
    close(dz, z.grad, rtol=1e-4, name="se dz")


def test_asp(ops):
    B, C, T = 2, 96, 83
    x = F.relu(synth_feat((B, C, T), 1)).double().requires_grad_(True)
    a = synth_feat((B, C, T), 2).double().requires_grad_(True)
    w = torch.softmax(a, dim=2)
    mu = torch.sum(x * w, dim=2)
    sg = torch.sqrt((torch.sum((x ** 2) * w, dim=2) - mu ** 2).clamp(min=1e-4))
    out = torch.cat((mu, sg), 1)
    dout = synth_feat((B, 2 * C), 3)
    out.backward(dout.double())
    xg = x.detach().float().cuda()
    wg = a.detach().float().cuda().clone()
    got = ops.asp_fwd(xg, wg)
    close(got, out, name="asp fwd")
    close(wg, w, name="asp weights")
    dx = torch.empty(B, C, T, device="cuda")
    ops.asp_bwd(xg, wg, got, dout.cuda(), dx, accumulate=False)
    close(dx, x.grad, rtol=1e-4, name="asp dx")
    close(wg, a.grad, rtol=1e-4, name="asp dlogits")


def test_bn_after_relu_backward(ops):
    """conv -> ReLU -> BN ordering (ecapa_tdnn.py:67-69): bn_bwd(relu_in) returns d(pre-ReLU)."""
    B, C, T = 4, 64, 50
    c = synth_feat((B, C, T), 1).double().requires_grad_(True)
    gamma = (1 + 0.2 * synth_feat((C,), 2)).double()
    beta = (0.1 * synth_feat((C,), 3)).double()
    r = F.relu(c)
    y = F.batch_norm(r, None, None, gamma, beta, True, 0.1, 1e-5)
    dy = synth_feat((B, C, T), 4)
    y.backward(dy.double())
    rg = r.detach().float().cuda()
    mean, invstd, scale, shift = ops.bn_stats(rg, gamma.float().cuda(), beta.float().cuda())
    close(ops.bn_apply(rg, scale, shift, False), y, name="bn fwd")
    dc, _, _ = ops.bn_bwd(rg, dy.cuda(), mean, invstd, gamma.float().cuda(), beta.float().cuda(),
                          relu=False, relu_in=True)
    close(dc, c.grad, rtol=1e-4, name="d pre-relu")
    d2 = dy.cuda().clone()
    ops.relu_mask_(d2, rg)
    close(d2, dy.double() * (r > 0), name="relu mask")
    lin = ops.linear_fwd(synth_feat((3, 40), 5).cuda(), synth_feat((7, 40), 6).cuda(), synth_feat((7,), 7).cuda(), relu=True)
    close(lin, F.relu(F.linear(synth_feat((3, 40), 5), synth_feat((7, 40), 6), synth_feat((7,), 7))), name="linear relu")


@pytest.mark.parametrize("B,C,T", [(4, 64, 50), (6, 512, 75), (3, 128, 750)])
def test_bn_backward_fused_bias_and_rowbias(ops, B, C, T):
    """air_bn_bwd_ex: (a) the conv-bias gradient sum_{b,t} d(pre-ReLU) comes out of the statistics pass in
    closed form; (b) a per-(b, c) constant added to the incoming gradient (the SE squeeze's 1/T term,
    ecapa_tdnn.py:19) is applied inside both passes.  Reference: fp64 autograd of
    y = BN(relu(conv_out + bias)), loss = <y, dy> + <mean_T(y), dm>."""
    c = synth_feat((B, C, T), 1).double().requires_grad_(True)
    bias = (0.3 * synth_feat((C,), 9)).double().requires_grad_(True)
    gamma = (1 + 0.2 * synth_feat((C,), 2)).double()
    beta = (0.1 * synth_feat((C,), 3)).double()
    r = F.relu(c + bias[None, :, None])
    y = F.batch_norm(r, None, None, gamma, beta, True, 0.1, 1e-5)
    dy = synth_feat((B, C, T), 4)
    dm = synth_feat((B, C), 5)
    ((y * dy.double()).sum() + (y.mean(2) * dm.double()).sum()).backward()
    rg = r.detach().float().cuda()
    mean, invstd, _, _ = ops.bn_stats(rg, gamma.float().cuda(), beta.float().cuda())
    dbias = torch.empty(C, device="cuda")
    dc, dg, db = ops.bn_bwd(rg, dy.cuda(), mean, invstd, gamma.float().cuda(), beta.float().cuda(), relu=False,
                            relu_in=True, rowbias=dm.cuda(), rowbias_scale=1.0 / T, dbias=dbias)
    close(dc, c.grad, rtol=1e-4, name="d pre-relu with row bias")
    close(dbias, bias.grad, rtol=1e-4, name="conv bias gradient (closed form)")
    close(dbias, dc.sum(dim=(0, 2)).double(), rtol=1e-4, name="== channel sum of dx")
    # in place (dx aliases dy), as the model calls it
    d2 = dy.cuda().clone()
    ops.bn_bwd(rg, d2, mean, invstd, gamma.float().cuda(), beta.float().cuda(), relu=False, relu_in=True, dx=d2,
               rowbias=dm.cuda(), rowbias_scale=1.0 / T, dbias=dbias)
    close(d2, c.grad, rtol=1e-4, name="in place")
    # the incoming gradient as the sum of two channel-slice views (the Res2 chain's join)
    big1 = synth_feat((B, 3 * C, T), 11).cuda()
    big2 = synth_feat((B, 2 * C, T), 12).cuda()
    part = dy.cuda() - big2[:, C:]
    big1[:, C:2 * C] = part
    dx3 = torch.empty(B, C, T, device="cuda")
    ops.bn_bwd(rg, big1[:, C:2 * C], mean, invstd, gamma.float().cuda(), beta.float().cuda(), relu=False, relu_in=True,
               dx=dx3, dy2=big2[:, C:], rowbias=dm.cuda(), rowbias_scale=1.0 / T, dbias=dbias)
    close(dx3, c.grad, rtol=1e-4, name="two strided gradients joined")
    close(dbias, bias.grad, rtol=1e-4, name="bias gradient, joined gradients")
