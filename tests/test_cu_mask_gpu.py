"""The persistent Winograd kernels on a device that does not offer them all of its CUs.

BASELINE configs[3] overlaps the RCCL all-reduce with the backward pass: collective kernels are resident on some
CUs while ``wino4_conv_kernel`` - one persistent workgroup per CU, 144 KB of LDS, tail items cut between two
workgroups that meet through a flag - is launched.  The same happens on a partitioned device (CPX) or under a CU
mask.  A stream created with ``hipExtStreamCreateWithCUMask`` (192 of the 256 CUs) is that situation on one GPU:
the library must size its grid by the CUs the stream can use, every cut item must still find its partner (no
trap), and the results must be the unmasked launch's up to the summation order of the cut items.
"""
import ctypes

import numpy as np

import pytest
import torch

from asvspoof2021_air_amd import _hip, ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def masked_stream():
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        pytest.skip("libamdhip64 not loadable through ctypes")
    total = torch.cuda.get_device_properties(0).multi_processor_count
    keep = total * 3 // 4
    words = (total + 31) // 32
    bits = [0] * words
    for cu in range(keep):
        bits[cu // 32] |= 1 << (cu % 32)
    mask = (ctypes.c_uint32 * words)(*bits)
    handle = ctypes.c_void_p()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
    if rc != 0 or not handle.value:
        pytest.skip("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    yield torch.cuda.ExternalStream(handle.value), keep, total
    torch.cuda.synchronize()
    hip.hipStreamDestroy(handle)


def test_stream_compute_units(masked_stream):
    ext, keep, total = masked_stream
    L = _hip.lib()
    assert L.air_stream_compute_units(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == total
    assert L.air_stream_compute_units(ctypes.c_void_p(ext.cuda_stream)) == keep


# (Cin, H, W, Cout) of the ResNet's 3x3 / stride 1 layers (resnet.py:56-61) at a batch whose item count
# leaves a half-empty last round on BOTH grids, so tail items are cut and meet through the flag
@pytest.mark.parametrize("shape", [(64, 18, 750, 64), (128, 9, 375, 128), (256, 5, 188, 256), (512, 3, 94, 512)])
def test_wino4_on_masked_stream_matches_full_chip(masked_stream, shape):
    ext, keep, total = masked_stream
    Cin, H, W, Cout = shape
    B = 20
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn((B, Cin, H, W), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (3.0 * Cin ** 0.5)).cuda()
    res = torch.randn((B, Cout, H, W), generator=g).cuda()
    dy = torch.randn((B, Cout, H, W), generator=g).cuda()
    want_f = ops.conv2d_fwd(x, w, 1, 1, residual=res)
    want_d = ops.conv2d_dgrad(dy, w, x.shape, 1, 1)
    torch.cuda.synchronize()
    ext.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(ext):
        for _ in range(3):  # repeated launches reuse the flag ring's slots
            got_f = ops.conv2d_fwd(x, w, 1, 1, residual=res)
            got_d = ops.conv2d_dgrad(dy, w, x.shape, 1, 1)
    ext.synchronize()  # a trapped launch (partner never ran) surfaces here as a HIP error
    for got, want in ((got_f, want_f), (got_d, want_d)):
        scale = float(want.abs().max())
        assert torch.isfinite(got).all()
        # same kernel, same arithmetic per item; only the items cut in two k-halves differ in summation order
        assert float((got - want).abs().max()) <= 1.52e-5 * scale


def test_resnet_train_step_on_masked_stream(masked_stream):
    """The whole ResNet step (forward, backward incl. the side-stream weight gradients, Adam) issued from the
    masked stream: same loss and gradients as on the full chip within the convolution's rounding budget."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    from oracle.filler import fill_module_, synth_pcm
    ext, keep, total = masked_stream
    pcm = synth_pcm(12, 32000, seed=5).cuda()
    labels = (torch.arange(12) % 3 != 0).long().cuda()
    outs = []
    for stream in (None, ext):
        torch.manual_seed(688)
        m = ResNet(3, 256, resnet_type="18", nclasses=2)
        fill_module_(m)
        m.set_attention_noise(None)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        tr = Trainer(m, loss_module=lossm, feat_len=201)
        torch.cuda.synchronize()
        if stream is None:
            torch.manual_seed(1)
            loss, _ = tr.step(pcm, labels)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                torch.manual_seed(1)
                loss, _ = tr.step(pcm, labels)
            stream.synchronize()
        torch.cuda.synchronize()
        outs.append((float(loss), m.arena().grad.clone()))
    (l0, g0), (l1, g1) = outs
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    rel = float((g0 - g1).norm() / g0.norm())
    assert rel <= 5e-3, rel


# ---------------------------------------------------------------------------------------------------------------
# a REAL kernel occupying CUs on another stream while full-grid persistent launches with cut
# items are dispatched (the masked stream above only shrinks the grid).  air_debug_cu_hog holds N compute units
# (N workgroups x 64 KB of LDS: a 144 KB wino4 workgroup cannot share the CU) for a few milliseconds - the position
# RCCL's all-reduce kernels are in during BASELINE configs[3]'s backward pass.  The grid is still one workgroup per CU
# of the device, so N of them cannot be resident until the hog leaves or an earlier workgroup finishes: the
# publisher / owner protocol of the cut items (conv_wino4.hip) must not depend on all workgroups running at once.
def _hog(stream, nblocks, ms):
    _hip.check(_hip.lib().air_debug_cu_hog(ctypes.c_int(nblocks), ctypes.c_int(64 * 1024), ctypes.c_double(ms),
                                           ctypes.c_void_p(stream.cuda_stream)), "air_debug_cu_hog")


@pytest.mark.parametrize("nhog", [16, 64])
@pytest.mark.parametrize("shape", [(64, 18, 750, 64), (256, 5, 188, 256), (512, 3, 94, 512)])
def test_wino4_with_resident_hog_kernel(shape, nhog):
    Cin, H, W, Cout = shape
    B = 20
    g = torch.Generator().manual_seed(Cin + H + 1)
    x = torch.randn((B, Cin, H, W), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / (3.0 * Cin ** 0.5)).cuda()
    res = torch.randn((B, Cout, H, W), generator=g).cuda()
    dy = torch.randn((B, Cout, H, W), generator=g).cuda()
    main = torch.cuda.current_stream()
    hog_stream = torch.cuda.Stream()

    def run(n):
        outs = []
        for _ in range(n):
            outs.append((ops.conv2d_fwd(x, w, 1, 1, residual=res, stats=True), ops.conv2d_dgrad(dy, w, x.shape, 1, 1)))
        return outs

    run(2)  # warm: lazy attributes, the flag ring
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main)
    want = run(4)
    e1.record(main)
    torch.cuda.synchronize()
    free_ms = e0.elapsed_time(e1)
    hog_ms = 6.0
    _hog(hog_stream, nhog, hog_ms)       # resident first ...
    e1.record(main)
    got = run(4)                         # ... then the full-grid launches, dispatched while it holds its CUs
    e2.record(main)
    torch.cuda.synchronize()             # a hang (owner spinning on a never-dispatched publisher) would time out here
    hogged_ms = e1.elapsed_time(e2)
    for ((yf, rec), dx), ((yf0, rec0), dx0) in zip(got, want):
        assert torch.equal(yf, yf0) and torch.equal(dx, dx0)          # same dealing, same summation order: bit-equal
        assert (rec is None) == (rec0 is None) and (rec is None or torch.equal(rec, rec0))
    # bounded slowdown: at worst the launches wait for the hog to leave (its 6 ms) and then run as usual
    assert hogged_ms <= free_ms * 1.5 + hog_ms + 1.0, (free_ms, hogged_ms)


def test_resnet_train_step_with_resident_hog_kernel():
    """The whole ResNet step (side-stream weight gradients included) while 32 CUs are held by another kernel for most
    of it: bit-identical loss and gradients."""
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.train import Trainer
    from oracle.filler import fill_module_, synth_pcm
    pcm = synth_pcm(12, 32000, seed=5).cuda()
    labels = (torch.arange(12) % 3 != 0).long().cuda()
    hog_stream = torch.cuda.Stream()
    outs = []
    for hog in (False, True):
        m = ResNet(3, 256, resnet_type="18", nclasses=2)
        fill_module_(m)
        m.set_attention_noise(None)
        lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossm)
        tr = Trainer(m, loss_module=lossm, feat_len=201)
        tr.step(pcm, labels)  # warm-up step (also gives the second step its prepacked weights)
        torch.cuda.synchronize()
        if hog:
            for _ in range(3):
                _hog(hog_stream, 32, 8.0)
        loss, _ = tr.step(pcm, labels)
        torch.cuda.synchronize()
        outs.append((float(loss), m.arena().grad.clone(), m.arena().flat.clone()))
    (l0, g0, w0), (l1, g1, w1) = outs
    assert l0 == l1 and torch.equal(g0, g1) and torch.equal(w0, w1)


@pytest.mark.gpu
def test_clock_probe_beside_a_kernel():
    """air_debug_clock_probe (bench.py timing.core_clock_mhz_during_steps): one wave on a side stream samples {wall clock,
    core-clock counter} while another kernel runs; the samples are monotonic, evenly spaced and give a plausible clock."""
    from asvspoof2021_air_amd.ops import ClockProbe
    dev = torch.device("cuda")
    probe = ClockProbe(dev, n_samples=200, interval_us=50.0)
    probe.start()
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(10):
        a = a * 1.0001 + 0.5
    torch.cuda.current_stream().synchronize()
    t, f = probe.samples()
    assert len(t) >= 150 and bool((np.diff(t) > 0).all())
    assert abs(float(np.median(np.diff(t))) - 50.0) < 10.0
    r = probe.mhz()
    assert 300.0 < r["min"] <= r["median"] <= r["max"] < 3500.0
    with pytest.raises(Exception):
        ClockProbe(dev, n_samples=1).start()
