"""CPU: the C-ABI shared library builds, loads and exports every symbol that
include/air_hip.h declares; host-only entry points behave (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from asvspoof2021_air_amd import _hip, build
    build.build(verbose=False)
    return _hip.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "air_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(air_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in air_hip.h but not exported: %s" % missing


def test_version_and_sizes(lib):
    assert lib.air_version().startswith(b"air_hip gfx950")
    assert lib.air_abi_version() >= 1
    assert lib.air_lfcc_plan_bytes() > 0
    assert lib.air_preemph_ws_bytes(ctypes.c_int(2), ctypes.c_int(64000)) == 2 * 16 * 4
    assert lib.air_preemph_ws_bytes(ctypes.c_int(0), ctypes.c_int(64000)) == 0


def test_dispatch_options(lib):
    """One setter for every dispatch switch (include/air_hip.h): names listed, defaults, round trip, and every
    option the header documents exists."""
    from asvspoof2021_air_amd import _hip
    names = [lib.air_option_name(ctypes.c_int(i)).decode() for i in range(lib.air_option_count())]
    assert lib.air_option_name(ctypes.c_int(len(names))) is None
    text = open(os.path.join(ROOT, "include", "air_hip.h")).read()
    for n in names:
        assert re.search(r"\b%s\b" % n, text), "option %s is not documented in air_hip.h" % n
    assert "NO_WINO4" in names and "NO_WINOGRAD" in names
    assert _hip.get_option("WGRAD_WGS") == int(os.environ.get("AIR_WGRAD_WGS", 256))
    with _hip.options(NO_WINOGRAD=3, AIR_NO_WINO4=1):  # the AIR_ prefix is accepted too
        assert _hip.get_option("NO_WINOGRAD") == 3 and _hip.get_option("NO_WINO4") == 1
    assert _hip.get_option("NO_WINOGRAD") == int(os.environ.get("AIR_NO_WINOGRAD", 0))
    assert lib.air_set_option(b"NOT_AN_OPTION", ctypes.c_int(1)) == -1
    assert lib.air_get_option(b"NO_WINO4", None) == -1


def test_lfcc_plan_build_host(lib, golden):
    from asvspoof2021_air_amd import _hip
    g = golden("lfcc.npz")
    fb = torch.from_numpy(g["fb"].copy())
    dct = torch.from_numpy(g["dct"].copy())
    plan = torch.zeros(lib.air_lfcc_plan_bytes(), dtype=torch.uint8)
    rc = lib.air_lfcc_plan_build(_hip.hptr(fb), 257, 20, _hip.hptr(dct), None, 320, 160, 512,
                                 _hip.hptr(plan, torch.uint8))
    assert rc == 0
    hdr = plan[:32].view(torch.int32)
    assert hdr[0] == 20 and hdr[1] == 25  # 20 filters, widest spans 25 bins (486 non-zeros in all)
    win = plan[32:32 + 320 * 4].view(torch.float32).numpy()
    np.testing.assert_allclose(win, 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(320) / 320), atol=1e-7)
    # unsupported geometry / bad arguments are reported, not crashed on
    assert lib.air_lfcc_plan_build(_hip.hptr(fb), 257, 20, _hip.hptr(dct), None, 400, 160, 512,
                                   _hip.hptr(plan, torch.uint8)) == -2
    assert lib.air_lfcc_plan_build(None, 257, 20, _hip.hptr(dct), None, 320, 160, 512,
                                   _hip.hptr(plan, torch.uint8)) == -1
    wide = torch.ones(257, 20)
    assert lib.air_lfcc_plan_build(_hip.hptr(wide), 257, 20, _hip.hptr(dct), None, 320, 160, 512,
                                   _hip.hptr(plan, torch.uint8)) == -2  # filter wider than the sparse table


def test_conv_workspace_queries(lib):
    from asvspoof2021_air_amd._hip import AirConv2d
    ok = AirConv2d(64, 64, 18, 750, 64, 3, 3, 1, 1, 1, 1, 18, 750)
    assert lib.air_conv2d_ws_bytes(ctypes.byref(ok)) > 64 * 64 * 9 * 4
    bad_shape = AirConv2d(64, 64, 18, 750, 64, 3, 3, 1, 1, 1, 1, 18, 751)
    assert lib.air_conv2d_ws_bytes(ctypes.byref(bad_shape)) == 0
    unsupported = AirConv2d(1, 24, 8, 8, 24, 5, 5, 1, 1, 2, 2, 8, 8)
    assert lib.air_conv2d_ws_bytes(ctypes.byref(unsupported)) == 0
    assert lib.air_bn_ws_bytes(ctypes.c_int(64), ctypes.c_int(64), ctypes.c_int(13500)) > 0


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback anywhere on the product path."""
    from asvspoof2021_air_amd import _hip
    from asvspoof2021_air_amd.feature_extraction import LFCC
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    with pytest.raises(_hip.AirError):
        LFCC(320, 160, 512, 16000, 20)(torch.zeros(1, 1600))
    with pytest.raises(_hip.AirError):
        ResNet(3, 256)(torch.zeros(2, 1, 60, 96))
    with pytest.raises(_hip.AirError):
        AngularIsoLoss(256)(torch.zeros(2, 256), torch.zeros(2, dtype=torch.long))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from asvspoof2021_air_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.HipExtensionMissing):
        _hip.lib()


def test_module_surface_matches_reference():
    """Constructor signatures / state_dict keys of the drop-in modules."""
    import inspect
    from asvspoof2021_air_amd.feature_extraction import LFCC
    from asvspoof2021_air_amd.loss import AngularIsoLoss, OCSoftmax
    from asvspoof2021_air_amd.resnet import ResNet
    from oracle import resnet as o_resnet
    assert list(inspect.signature(LFCC.__init__).parameters)[1:] == [
        "fl", "fs", "fn", "sr", "filter_num", "with_energy", "with_emphasis", "with_delta"]
    assert list(inspect.signature(ResNet.__init__).parameters)[1:] == ["num_nodes", "enc_dim", "resnet_type", "nclasses"]
    sig = inspect.signature(AngularIsoLoss.__init__).parameters
    assert [sig[k].default for k in ("feat_dim", "r_real", "r_fake", "alpha")] == [2, 0.9, 0.5, 20.0]
    assert issubclass(OCSoftmax, AngularIsoLoss)
    m = ResNet(3, 256, resnet_type="18", nclasses=2)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [
        (k, tuple(v)) for k, v in o_resnet.resnet18_shapes().items()]
    assert list(LFCC(320, 160, 512, 16000, 20).state_dict()) == ["lfcc_fb", "l_dct.weight"]
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    assert tuple(lossm.center.shape) == (1, 256) and lossm.center.abs().max() <= 0.15  # kaiming_uniform(a=0.25)


def test_product_eer_matches_reference_goldens(golden):
    from asvspoof2021_air_amd.eval_metrics import compute_eer
    g = golden("eer.npz")
    e1, t1 = compute_eer(g["tgt"], g["non"])
    e2, t2 = compute_eer(g["tgt_t"], g["non_t"])
    np.testing.assert_allclose([e1, e2], g["eer"], atol=1e-12)
    np.testing.assert_allclose([t1, t2], g["thr"], atol=1e-12)
