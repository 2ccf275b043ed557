"""Channel augmentation: oracle self-checks on CPU, HIP FIR kernel vs the oracle on GPU.
Parity with the reference's external tool is UNPINNED (oracle/channel.py header)."""
import numpy as np
import pytest
import torch

from oracle import channel as o_channel
from oracle.filler import synth_pcm


def test_oracle_fftconvolve_equals_definition():
    rng = np.random.default_rng(3)
    x, h = rng.standard_normal(300), rng.standard_normal(37)
    want = o_channel.ir_convolve_direct(x, h)
    got = o_channel.ir_convolve(x[None], h[None], [0], normalize=False)[0]
    np.testing.assert_allclose(got, want, atol=1e-12)
    norm = o_channel.ir_convolve(x[None], h[None], [0], normalize=True)[0]
    np.testing.assert_allclose(np.abs(norm).max(), np.abs(x).max(), rtol=1e-12)
    np.testing.assert_allclose(norm / np.abs(norm).max(), want / np.abs(want).max(), atol=1e-12)
    assert np.array_equal(o_channel.ir_convolve(x[None], h[None], [-1])[0], x)  # pass-through


def test_synthetic_bank_is_deterministic_and_unit_energy():
    from asvspoof2021_air_amd.augment import synthetic_ir_bank
    a, b = synthetic_ir_bank(seed=5), synthetic_ir_bank(seed=5)
    assert a.shape == (30, 1024) and torch.equal(a, b)
    np.testing.assert_allclose((a.double() ** 2).sum(1).numpy(), 1.0, rtol=1e-6)
    assert not torch.equal(a, synthetic_ir_bank(seed=6))


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,H", [(3, 64000, 1024), (2, 5000, 37), (2, 2049, 1), (2, 777, 2500), (1, 4096, 1025),
                                   # the overlap-save FFT form (128 <= H <= 1025): shortest / longest response, one block,
                                   # one sample into the second block of a pair, one sample into the second pair
                                   (2, 6149, 128), (3, 9300, 700), (2, 3072, 1024), (2, 3073, 1024), (2, 6145, 1025)])
def test_fir_kernel_vs_oracle(B, L, H):
    from asvspoof2021_air_amd.augment import ir_convolve
    rng = np.random.default_rng(L + H)
    x = synth_pcm(B, L, seed=9)
    irs = torch.from_numpy((rng.standard_normal((4, H)) * np.exp(-np.arange(H) / max(H / 6.0, 1.0))).astype(np.float32))
    idx = np.array([1, 3, -1, 0][:B], dtype=np.int32)
    for normalize in (False, True):
        got = ir_convolve(x.cuda(), irs.cuda(), torch.from_numpy(idx).cuda(), normalize).cpu().double().numpy()
        want = o_channel.ir_convolve(x.numpy(), irs.numpy(), idx, normalize)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-6 * scale * max(1.0, np.sqrt(H) / 8), (normalize, np.abs(got - want).max(), scale)
        for b in range(B):
            if idx[b] < 0:
                assert np.array_equal(got[b], x[b].double().numpy())  # untouched, bit-exact
    one = torch.zeros(1, 1)
    one[0, 0] = 1.0
    same = ir_convolve(x.cuda(), one.cuda(), None, False).cpu()
    assert torch.equal(same, x)  # h = delta: identity, bit-exact


@pytest.mark.gpu
@pytest.mark.parametrize("L,H", [(64000, 1024), (10000, 300)])
def test_fft_form_equals_direct_form(L, H):
    """Option IR_FFT: the overlap-save FFT kernel and the direct FIR kernel on the same input - equal to fp32 FFT rounding
    (4e-6 of the output scale), pass-through utterances bit-identical in both, the rescaled peaks equal to 1e-6."""
    from asvspoof2021_air_amd import _hip
    from asvspoof2021_air_amd.augment import ir_convolve
    rng = np.random.default_rng(L)
    x = synth_pcm(5, L, seed=3).cuda()
    irs = torch.from_numpy((rng.standard_normal((3, H)) * np.exp(-np.arange(H) / (H / 6.0))).astype(np.float32)).cuda()
    idx = torch.tensor([0, 2, -1, 1, 2], dtype=torch.int32).cuda()
    out = {}
    for mode in (1, 0):
        old = _hip.set_option("IR_FFT", mode)
        try:
            out[mode] = [ir_convolve(x, irs, idx, nz).cpu() for nz in (False, True)]
        finally:
            _hip.set_option("IR_FFT", old)
    for a, b in zip(out[1], out[0]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 4e-6 * scale
        assert torch.equal(a[2], x[2].cpu()) and torch.equal(b[2], x[2].cpu())
    assert float((out[1][1].abs().amax(1) - out[0][1].abs().amax(1)).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_channel_augment_in_front_end():
    """Seeded per-utterance choice; the augmented PCM then goes through the fused LFCC kernel and
    matches the oracle LFCC of the oracle-augmented PCM."""
    from asvspoof2021_air_amd.augment import ChannelAugment
    from asvspoof2021_air_amd.feature_extraction import LFCC
    from oracle import lfcc as o_lfcc
    aug = ChannelAugment(p=0.5, seed=1)
    idx = aug.draw(8)
    assert np.array_equal(idx, ChannelAugment(p=0.5, seed=1).draw(8))
    assert (idx < 0).any() and (idx >= 0).any() and idx.max() < 30
    x = synth_pcm(8, 16000, seed=2)
    y = aug(x.cuda(), idx)
    want = o_channel.ir_convolve(x.numpy(), aug.irs.cpu().numpy(), idx, True)
    np.testing.assert_allclose(y.cpu().numpy(), want, atol=2e-6)
    lf = LFCC(320, 160, 512, 16000, 20, with_energy=False).cuda()
    lf.mutate_input = False
    feat = lf(y).cpu().numpy()
    ref = o_lfcc.lfcc_forward(want.astype(np.float32))
    np.testing.assert_allclose(feat, ref, atol=2e-3)  # log10 of near-empty bins amplifies the 1e-6 PCM difference
