"""GPU parity of the bf16 pointwise-Conv1d kernels (csrc/conv1d_bf16.hip, BASELINE configs[2]).

The kernels round both operands to bf16 (nearest even) and accumulate exact products in fp32,
so the reference is an fp64 convolution of the bf16-ROUNDED operands: what remains is fp32
summation order.  Tolerance 2e-5 of the output scale, the same as the fp32 kernels; against the
UNROUNDED fp64 result the error is the bf16 operand rounding (checked to be in its 2^-9 class)."""
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


def bf(t):
    return t.float().bfloat16().double()


def relerr(got, want):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    assert got.shape == want.shape
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


CASES = [  # (B, Cin, T, Cout)
    (3, 512, 75, 512),     # Bottle2neck conv1 / conv3, ragged T (one partial time tile)
    (2, 1536, 40, 1536),   # layer4
    (2, 1536, 40, 128),    # attention.0 (x part)
    (2, 128, 40, 1536),    # attention.3
    (2, 512, 750, 512),    # reference feat_len: 6 time tiles, last one 110 wide; 24 wgrad stages per utterance
    (5, 128, 401, 128),    # odd T (native 4 s frame count): scalar-load path of the wgrad kernel
    (1, 128, 1, 128),      # single frame
    (2, 512, 40, 2048),    # asymmetric wide layers: forward and dgrad both on the GEMM path, transposed
    (2, 2048, 40, 512),    # operand copies of different widths (workspace sized for the larger)
]


@pytest.mark.parametrize("cfg", CASES)
def test_conv1d_bf16(ops, cfg):
    B, Cin, T, Cout = cfg
    x = synth_feat((B, Cin, T), 1)
    w = synth_feat((Cout, Cin, 1), 2, scale=0.05)
    b = synth_feat((Cout,), 3, scale=0.2)
    bbc = synth_feat((B, Cout), 4, scale=0.2)
    dy = synth_feat((B, Cout, T), 5)
    acc = synth_feat((B, Cin, T), 6)
    xg, wg, dyg = x.cuda(), w.cuda(), dy.cuda()
    # forward
    pre = F.conv1d(bf(x), bf(w)) + b.double()[None, :, None] + bbc.double().unsqueeze(2)
    got = ops.conv1d_fwd(xg, wg, b.cuda(), bbc.cuda(), relu=True, bf16=True)
    assert relerr(got, F.relu(pre)) <= 2e-5
    exact = F.relu(F.conv1d(x.double(), w.double()) + b.double()[None, :, None] + bbc.double().unsqueeze(2))
    e = relerr(got, exact)
    assert 1e-5 < e < 2e-2 or Cin * T < 1000, "bf16 rounding class expected, got %.3g" % e
    got = ops.conv1d_fwd(xg, wg, bf16=True)
    assert relerr(got, F.conv1d(bf(x), bf(w))) <= 2e-5
    # dgrad (+ accumulate)
    want = F.conv_transpose1d(bf(dy), bf(w))
    assert relerr(ops.conv1d_dgrad(dyg, wg, bf16=True), want) <= 2e-5
    got = ops.conv1d_dgrad(dyg, wg, accumulate=acc.cuda(), bf16=True)
    assert relerr(got, want + acc.double()) <= 2e-5
    # wgrad
    want = torch.einsum("bot,bit->oi", bf(dy), bf(x)).unsqueeze(2)
    assert relerr(ops.conv1d_wgrad(xg, dyg, (Cout, Cin, 1), bf16=True), want) <= 2e-5


def test_conv1d_bf16_channel_slice_views(ops):
    """Batch-strided channel-slice operands and outputs (the (x1,x2,x3) concat of ecapa_tdnn.py:170)."""
    B, T = 3, 50
    big = synth_feat((B, 384, T), 7).cuda()
    x = big[:, 128:256]
    w = synth_feat((128, 128, 1), 8, scale=0.05)
    outbig = torch.full((B, 384, T), 7.0, device="cuda")
    ops.conv1d_fwd(x, w.cuda(), out=outbig[:, 256:], bf16=True)
    want = F.conv1d(bf(x.cpu()), bf(w))
    assert relerr(outbig[:, 256:], want) <= 2e-5
    assert float(outbig[:, :256].min()) == 7.0 and float(outbig[:, :256].max()) == 7.0
    dy = synth_feat((B, 384, T), 9).cuda()
    got = ops.conv1d_wgrad(x, dy[:, :128], (128, 128, 1), bf16=True)
    want = torch.einsum("bot,bit->oi", bf(dy[:, :128].cpu()), bf(x.cpu())).unsqueeze(2)
    assert relerr(got, want) <= 2e-5
    dx = torch.zeros((B, 384, T), device="cuda")
    ops.conv1d_dgrad(dy[:, :128], w.cuda(), out=dx[:, 128:256], bf16=True)
    assert relerr(dx[:, 128:256], F.conv_transpose1d(bf(dy[:, :128].cpu()), bf(w))) <= 2e-5
    assert float(dx[:, :128].abs().max()) == 0.0


@pytest.mark.parametrize("B,C,T,dil", [(2, 64, 100, 2), (3, 64, 750, 3), (2, 64, 33, 4), (2, 128, 129, 4), (1, 64, 3, 2)])
def test_conv1d_bf16_dilated_k3(ops, B, C, T, dil):
    """The Res2 branch convs (ecapa_tdnn.py:46): forward (+bias, ReLU) and dgrad (+accumulate) in bf16 compute
    against fp64 contractions of the bf16-rounded operands; zero padding, halo across time tiles, T < halo."""
    x = synth_feat((B, C, T), 1)
    w = synth_feat((C, C, 3), 2, scale=0.05)
    b = synth_feat((C,), 3, scale=0.2)
    dy = synth_feat((B, C, T), 5)
    acc = synth_feat((B, C, T), 6)
    want = F.relu(F.conv1d(bf(x), bf(w), None, 1, dil, dil) + b.double()[None, :, None])
    got = ops.conv1d_fwd(x.cuda(), w.cuda(), b.cuda(), relu=True, dil=dil, pad=dil, bf16=True)
    assert relerr(got, want) <= 2e-5
    want = F.conv_transpose1d(bf(dy), bf(w), None, 1, dil, 0, 1, dil)
    got = ops.conv1d_dgrad(dy.cuda(), w.cuda(), dil, dil, bf16=True)
    assert relerr(got, want) <= 2e-5
    got = ops.conv1d_dgrad(dy.cuda(), w.cuda(), dil, dil, accumulate=acc.cuda(), bf16=True)
    assert relerr(got, want + acc.double()) <= 2e-5
    # strided views in and out (the Res2 groups are channel slices of wider tensors)
    big = synth_feat((B, 3 * C, T), 7).cuda()
    outbig = torch.zeros((B, 2 * C, T), device="cuda")
    ops.conv1d_fwd(big[:, C:2 * C], w.cuda(), dil=dil, pad=dil, out=outbig[:, C:], bf16=True)
    assert relerr(outbig[:, C:], F.conv1d(bf(big[:, C:2 * C].cpu()), bf(w), None, 1, dil, dil)) <= 2e-5
    assert float(outbig[:, :C].abs().max()) == 0.0
    # the weight gradient of these layers stays on the fp32 kernels
    wg = ops.conv1d_wgrad(x.cuda(), dy.cuda(), (C, C, 3), dil, dil, bf16=True)
    assert relerr(wg, torch.nn.grad.conv1d_weight(x.double(), (C, C, 3), dy.double(), 1, dil, dil)) <= 2e-5


def test_conv1d_bf16_falls_through_to_fp32_for_other_layers(ops):
    """K = 5 / ragged-channel layers are not the bf16 kernels': bf16=True runs the fp32 path."""
    x = synth_feat((2, 60, 100), 1)
    w = synth_feat((512, 60, 5), 2, scale=0.05)
    got = ops.conv1d_fwd(x.cuda(), w.cuda(), pad=2, bf16=True)
    assert relerr(got, F.conv1d(x.double(), w.double(), None, 1, 2, 1)) <= 2e-5


def test_conv1d_bf16_wgrad_is_deterministic(ops):
    x = synth_feat((16, 512, 200), 1).cuda()
    dy = synth_feat((16, 512, 200), 2).cuda()
    a = ops.conv1d_wgrad(x, dy, (512, 512, 1), bf16=True).clone()
    b = ops.conv1d_wgrad(x, dy, (512, 512, 1), bf16=True)
    assert torch.equal(a, b)
