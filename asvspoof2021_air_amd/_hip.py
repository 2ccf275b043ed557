"""ctypes binding of the C-ABI in include/air_hip.h (libair_hip.so).

This is the only way the package reaches the GPU kernels.  There is NO CPU or
PyTorch fallback: if the library is missing or a tensor is not on the GPU the
call raises.
"""
import ctypes
import os

import torch

# AIR_HIP_LIB: another build of the same library (A/B variants from ``build.py --variant``); never a fallback
_LIB_PATH = os.environ.get("AIR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libair_hip.so")
_lib = None

ERRORS = {-1: "AIR_EINVAL", -2: "AIR_EUNSUPPORTED", -3: "AIR_ELAUNCH", -4: "AIR_EWORKSPACE"}


class HipExtensionMissing(RuntimeError):
    pass


class AirError(RuntimeError):
    pass


class AirConv2d(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("B", "Cin", "H", "W", "Cout", "KH", "KW", "sh", "sw", "ph", "pw", "Ho", "Wo")]


class AirConv1d(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("B", "Cin", "T", "Cout", "K", "dil", "pad")] + [
        ("x_bstride", ctypes.c_size_t), ("y_bstride", ctypes.c_size_t)]


def lib_path():
    return _LIB_PATH


def lib():
    """Load libair_hip.so once.  Raises HipExtensionMissing (never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipExtensionMissing(
                "HIP extension not built: %s is missing. Run `python -m asvspoof2021_air_amd.build` "
                "(or __graft_entry__.build()). There is no CPU fallback." % _LIB_PATH)
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.air_version.restype = ctypes.c_char_p
        _lib.air_option_name.restype = ctypes.c_char_p
        _lib.air_lfcc_plan_bytes.restype = ctypes.c_size_t
        _lib.air_preemph_ws_bytes.restype = ctypes.c_size_t
        for name in ("air_conv2d_ws_bytes", "air_bn_ws_bytes", "air_conv1d_ws_bytes", "air_conv1d_bf16_ws_bytes",
                     "air_channel_sum_ws_bytes", "air_ir_convolve_ws_bytes", "air_ir_convolve_ws_bytes_ex", "air_conv2d_prepack_bytes",
                     "air_conv1d_tap_pack_elems", "air_h_bn_ws_bytes", "air_h_conv1d_ws_bytes", "air_conv2d_fwd_stats_bytes", "air_conv2d_dgrad_bn_sums_bytes", "air_conv2d_dgrad_s2_pair_prepack_bytes", "air_h_conv1d_tap_stats_bytes", "air_h_conv1d_tap_bwd_sums_bytes", "air_h_conv1d_pointwise_stats_bytes",
                     "air_h_conv1d_tap_wgrad_ws_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_size_t
    return _lib


def set_option(name, value):
    """Dispatch option of the library (include/air_hip.h, "dispatch options"); returns the previous value."""
    L = lib()
    old = ctypes.c_int(0)
    check(L.air_get_option(name.encode(), ctypes.byref(old)), "air_get_option(%s)" % name)
    check(L.air_set_option(name.encode(), ctypes.c_int(int(value))), "air_set_option(%s)" % name)
    return old.value


def get_option(name):
    v = ctypes.c_int(0)
    check(lib().air_get_option(name.encode(), ctypes.byref(v)), "air_get_option(%s)" % name)
    return v.value


class options:
    """``with _hip.options(NO_WINOGRAD=3): ...`` - set dispatch options for a block, restore them after."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def check(rc, what):
    if rc != 0:
        raise AirError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "?"), rc))


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def raw_stream():
    """The current HIP stream handle of the current device as an int.  torch.cuda.current_stream() builds a Stream
    object per call (7 us; ~350 calls per ECAPA step made the eager step host-bound); the raw getters cost 0.3 us."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def stream():
    return ctypes.c_void_p(raw_stream())


def dptr(t, dtype=torch.float32, allow_none=False):
    """Device pointer of a contiguous GPU tensor (or NULL)."""
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0)
        raise AirError("null tensor")
    if not t.is_cuda:
        raise AirError("tensor must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise AirError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise AirError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def hptr(t, dtype=torch.float32):
    """Host pointer of a contiguous CPU tensor."""
    if t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise AirError("expected contiguous CPU %s tensor" % dtype)
    return ctypes.c_void_p(t.data_ptr())


ci = ctypes.c_int
cf = ctypes.c_float
csz = ctypes.c_size_t
