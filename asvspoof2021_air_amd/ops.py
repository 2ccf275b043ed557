"""Tensor-level wrappers over the C-ABI (include/air_hip.h).

Each function takes contiguous fp32 GPU tensors, allocates outputs with torch
(device memory is plumbing) and launches the HIP kernels on the current
stream.  Nothing here computes on the CPU or through ATen.
"""
import ctypes

import torch

from . import _hip
from ._hip import AirConv1d, AirConv2d, ci, cf, csz, dptr, stream

_WS = {}
_WS_GEN = [0]     # bumped whenever a scratch buffer is (re)allocated
_WS_PINNED = {}   # capture owner -> buffers its hipGraph may still point into (kept alive until the capture is dropped)


def workspace_generation():
    """Changes whenever workspace() replaced a buffer.  A captured hipGraph holds raw pointers into the buffers
    that existed at capture time (train.Trainer compares this before every replay and re-captures)."""
    return _WS_GEN[0]


def pin_workspaces(owner=None):
    """Called when a hipGraph has been captured: from now on a buffer that is outgrown is kept alive instead of
    freed, so a replay can never write through a dangling scratch pointer - until ``unpin_workspaces(owner)``."""
    held = _WS_PINNED.setdefault(owner, [])
    for buf in _WS.values():
        if not any(buf is b for b in held):
            held.append(buf)


def unpin_workspaces(owner=None):
    """The capture that pinned the buffers is gone: release what was only kept alive for its replays."""
    _WS_PINNED.pop(owner, None)


def workspace(nbytes, device):
    """One growing scratch buffer per (device, stream): ops run back to back on a stream, and
    the weight-gradient side stream (resnet.py) must not share scratch with the main one."""
    cur = _hip._RAW_DEVICE() if _hip._RAW_DEVICE is not None else torch.cuda.current_device()
    key = (device.type, device.index,
           _hip.raw_stream() if device.index in (None, cur) else torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
        _WS_GEN[0] += 1
    return buf


def _conv_desc(x_shape, w_shape, stride, padding):
    B, Cin, H, W = x_shape
    Cout, Cin2, KH, KW = w_shape
    if Cin2 != Cin:
        raise _hip.AirError("conv2d: weight expects %d input channels, got %d" % (Cin2, Cin))
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    Ho = (H + 2 * ph - KH) // sh + 1
    Wo = (W + 2 * pw - KW) // sw + 1
    return AirConv2d(B, Cin, H, W, Cout, KH, KW, sh, sw, ph, pw, Ho, Wo)


def _conv_ws(desc, device):
    lib = _hip.lib()
    n = lib.air_conv2d_ws_bytes(ctypes.byref(desc))
    if n == 0:
        raise _hip.AirError("conv2d: unsupported configuration %s" %
                            ([getattr(desc, f[0]) for f in desc._fields_],))
    return workspace(n, device), n


def conv2d_prepack(w, x_shape, stride, padding, which, out=None):
    """Transformed (Winograd) weights of a 3x3 / stride 1 / pad 1 layer for ``which`` (0 forward, 1 dgrad) into a
    buffer the later conv2d_fwd / conv2d_dgrad call takes as ``w_packed``; None when the layer has no such form.
    Runs on the current stream: call it under a side stream and order it with events."""
    d = _conv_desc(x_shape, w.shape, stride, padding)
    n = int(_hip.lib().air_conv2d_prepack_bytes(ctypes.byref(d), ci(which)))
    if n == 0:
        return None
    if out is None or out.numel() < n:
        out = torch.empty(n, dtype=torch.uint8, device=w.device)
    _hip.check(_hip.lib().air_conv2d_prepack(ctypes.byref(d), dptr(w), ci(which), dptr(out, torch.uint8), csz(n),
                                             stream()), "air_conv2d_prepack")
    out._air_pack_layout = int(_hip.lib().air_conv2d_prepack_layout(ctypes.byref(d), ci(which)))
    return out


class prepack_batch:
    """``with ops.prepack_batch():`` the conv2d_prepack / conv2d_dgrad_s2_pair_prepack calls inside only record their
    transforms; leaving the block runs them on the current stream as one launch per 32 (air_conv2d_prepack_begin /
    _flush).  The buffers are valid behind the block in stream order."""

    def __enter__(self):
        _hip.check(_hip.lib().air_conv2d_prepack_begin(), "air_conv2d_prepack_begin")
        return self

    def __exit__(self, *exc):
        _hip.check(_hip.lib().air_conv2d_prepack_flush(stream()), "air_conv2d_prepack_flush")
        return False


def _check_packed(d, w_packed, which):
    """A prepacked weight buffer must be at least what the layer consumes under the CURRENT dispatch options
    (options can change between prepack and use; the library walks the buffer without a size)."""
    if w_packed is not None:
        #  the layouts differ in element type, not only in size: the tag conv2d_prepack left on the buffer
        # must be what the layer consumes now (f32 slabs / Winograd F2 / split-bf16 planes / Winograd F4)
        have = getattr(w_packed, "_air_pack_layout", None)
        now = int(_hip.lib().air_conv2d_prepack_layout(ctypes.byref(d), ci(which)))
        if have is not None and have != now:
            raise _hip.AirError("conv2d: w_packed was written in layout %d, the layer consumes layout %d under the current "
                                "dispatch options (prepack again after air_set_option)" % (have, now))
        need = int(_hip.lib().air_conv2d_prepack_bytes(ctypes.byref(d), ci(which)))
        if w_packed.numel() * w_packed.element_size() < need:
            raise _hip.AirError("conv2d: w_packed holds %d bytes, the layer consumes %d under the current dispatch "
                                "options (prepack again after air_set_option)" % (
                                    w_packed.numel() * w_packed.element_size(), need))


def conv2d_fwd(x, w, stride=1, padding=0, in_scale=None, in_shift=None, relu=False, residual=None, w_packed=None,
               stats=False):
    """y = conv2d(act(x), w) (+ residual); act = optional per-channel affine + ReLU.
    stats=True: returns (y, records) - BatchNorm statistics records of y from the convolution's epilogue for
    ``bn_stats(..., stats_in=records)``, or (y, None) when this layer has no fused statistics (the caller's BatchNorm
    then makes its own pass over y)."""
    d = _conv_desc(x.shape, w.shape, stride, padding)
    _check_packed(d, w_packed, 0)
    y = torch.empty((d.B, d.Cout, d.Ho, d.Wo), device=x.device, dtype=torch.float32)
    ws, n = _conv_ws(d, x.device)
    rec = None
    if stats and in_scale is None:
        nb = int(_hip.lib().air_conv2d_fwd_stats_bytes(ctypes.byref(d)))
        if nb > 0:
            rec = torch.empty(nb, dtype=torch.uint8, device=x.device)
    _hip.check(_hip.lib().air_conv2d_fwd_pre(
        ctypes.byref(d), dptr(x), dptr(w), dptr(w_packed, torch.uint8, allow_none=True), dptr(y),
        dptr(in_scale, allow_none=True),
        dptr(in_shift, allow_none=True), ci(1 if relu else 0), dptr(residual, allow_none=True),
        dptr(rec, torch.uint8, allow_none=True), dptr(ws, torch.uint8), csz(n), stream()), "air_conv2d_fwd_pre")
    return (y, rec) if stats else y


def conv2d_dgrad(dy, w, x_shape, stride=1, padding=0, accumulate=None, out=None, w_packed=None, bn=None):
    """dx = the data gradient of conv2d (+ accumulate).  bn = (bn_x, mean, invstd, gamma, beta): dx is the gradient
    with respect to the output of relu(batchnorm(bn_x)); returns (dx, sums) with ``sums`` the BatchNorm-backward
    records for ``bn_bwd(..., sums_in=sums)`` taken by the epilogue, or (dx, None) when this layer has no such form."""
    d = _conv_desc(x_shape, w.shape, stride, padding)
    _check_packed(d, w_packed, 1)
    dx = out if out is not None else torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    ws, n = _conv_ws(d, dy.device)
    if bn is not None:
        nb = int(_hip.lib().air_conv2d_dgrad_bn_sums_bytes(ctypes.byref(d)))
        if nb > 0:
            bx, mean, invstd, gamma, beta = bn
            if tuple(bx.shape) != tuple(x_shape):
                raise _hip.AirError("conv2d_dgrad: bn_x must have dx's shape")
            sums = torch.empty(nb, dtype=torch.uint8, device=dy.device)
            _hip.check(_hip.lib().air_conv2d_dgrad_bn(
                ctypes.byref(d), dptr(dy), dptr(w), dptr(w_packed, torch.uint8, allow_none=True), dptr(dx),
                dptr(accumulate, allow_none=True), dptr(bx), dptr(mean), dptr(invstd), dptr(gamma), dptr(beta),
                dptr(sums, torch.uint8), dptr(ws, torch.uint8), csz(n), stream()), "air_conv2d_dgrad_bn")
            return dx, sums
    _hip.check(_hip.lib().air_conv2d_dgrad_pre(
        ctypes.byref(d), dptr(dy), dptr(w), dptr(w_packed, torch.uint8, allow_none=True), dptr(dx),
        dptr(accumulate, allow_none=True),
        dptr(ws, torch.uint8), csz(n), stream()), "air_conv2d_dgrad_pre")
    return (dx, None) if bn is not None else dx


def conv2d_dgrad_s2_pair_ok(w_shape, x_shape):
    """Whether a 3x3 / stride 2 / pad 1 convolution and the 1x1 / stride 2 shortcut beside it have the one-pass data
    gradient under the current dispatch options."""
    d = _conv_desc(x_shape, w_shape, 2, 1)
    return int(_hip.lib().air_conv2d_dgrad_s2_pair_prepack_bytes(ctypes.byref(d))) > 0


def conv2d_dgrad_s2_pair_prepack(w, w_sc, x_shape, out=None):
    """Both weights of a PreActBlock's stride-2 pair (3x3 conv1, 1x1 shortcut) in the layout conv2d_dgrad_s2_pair
    consumes; None when the pair has no one-pass data gradient under the current dispatch options."""
    d = _conv_desc(x_shape, w.shape, 2, 1)
    n = int(_hip.lib().air_conv2d_dgrad_s2_pair_prepack_bytes(ctypes.byref(d)))
    if n == 0:
        return None
    if out is None or out.numel() < n:
        out = torch.empty(n, dtype=torch.uint8, device=w.device)
    _hip.check(_hip.lib().air_conv2d_dgrad_s2_pair_prepack(ctypes.byref(d), dptr(w), dptr(w_sc), dptr(out, torch.uint8),
                                                           csz(n), stream()), "air_conv2d_dgrad_s2_pair_prepack")
    return out


def conv2d_dgrad_s2_pair(dy, w, dy_sc, w_sc, x_shape, accumulate=None, out=None, packed=None):
    """dx = dgrad(3x3 / stride 2 / pad 1 conv; dy, w) + dgrad(1x1 / stride 2 shortcut; dy_sc, w_sc) (+ accumulate) in one
    pass over the two output gradients (resnet.py:56-66).  Returns None when the pair has no such form under the current
    dispatch options (the caller then runs the two data gradients one after the other)."""
    d = _conv_desc(x_shape, w.shape, 2, 1)
    if tuple(w_sc.shape) != (w.shape[0], w.shape[1], 1, 1) or tuple(dy_sc.shape) != tuple(dy.shape):
        raise _hip.AirError("conv2d_dgrad_s2_pair: the shortcut must be the 1x1 / stride 2 convolution beside the 3x3 one")
    n = int(_hip.lib().air_conv2d_dgrad_s2_pair_prepack_bytes(ctypes.byref(d)))
    if n == 0:
        return None
    if packed is not None and packed.numel() * packed.element_size() < n:
        raise _hip.AirError("conv2d_dgrad_s2_pair: packed holds %d bytes, the pair consumes %d under the current "
                            "dispatch options" % (packed.numel() * packed.element_size(), n))
    dx = out if out is not None else torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    ws = workspace(n, dy.device) if packed is None else None
    _hip.check(_hip.lib().air_conv2d_dgrad_s2_pair(
        ctypes.byref(d), dptr(dy), dptr(w), dptr(dy_sc), dptr(w_sc), dptr(packed, torch.uint8, allow_none=True), dptr(dx),
        dptr(accumulate, allow_none=True), dptr(ws, torch.uint8, allow_none=True), csz(n if packed is None else 0),
        stream()), "air_conv2d_dgrad_s2_pair")
    return dx


def conv2d_fwd_s2_pair_ok(w_shape, x_shape):
    """Whether a 3x3 / stride 2 / pad 1 convolution and the 1x1 / stride 2 shortcut on the same input have the one-launch
    forward under the current dispatch options."""
    d = _conv_desc(x_shape, w_shape, 2, 1)
    return int(_hip.lib().air_conv2d_fwd_s2_pair_prepack_bytes(ctypes.byref(d))) > 0


def conv2d_fwd_s2_pair_prepack(w, w_sc, x_shape, out=None):
    """Both weights of a PreActBlock's stride-2 pair in the layout conv2d_fwd_s2_pair consumes; None when the pair has
    no one-launch forward under the current dispatch options."""
    d = _conv_desc(x_shape, w.shape, 2, 1)
    n = int(_hip.lib().air_conv2d_fwd_s2_pair_prepack_bytes(ctypes.byref(d)))
    if n == 0:
        return None
    if out is None or out.numel() < n:
        out = torch.empty(n, dtype=torch.uint8, device=w.device)
    _hip.check(_hip.lib().air_conv2d_fwd_s2_pair_prepack(ctypes.byref(d), dptr(w), dptr(w_sc), dptr(out, torch.uint8),
                                                         csz(n), stream()), "air_conv2d_fwd_s2_pair_prepack")
    return out


def conv2d_fwd_s2_pair(x, w, w_sc, packed=None):
    """(conv2d(x, w, stride 2, pad 1), conv2d(x, w_sc, stride 2)) in one launch (resnet.py:56-66: both read the block's
    activated input); None when the pair has no such form under the current dispatch options."""
    d = _conv_desc(x.shape, w.shape, 2, 1)
    if tuple(w_sc.shape) != (w.shape[0], w.shape[1], 1, 1):
        raise _hip.AirError("conv2d_fwd_s2_pair: the shortcut must be the 1x1 / stride 2 convolution beside the 3x3 one")
    n = int(_hip.lib().air_conv2d_fwd_s2_pair_prepack_bytes(ctypes.byref(d)))
    if n == 0:
        return None
    if packed is not None and packed.numel() * packed.element_size() < n:
        raise _hip.AirError("conv2d_fwd_s2_pair: packed holds %d bytes, the pair consumes %d under the current dispatch "
                            "options" % (packed.numel() * packed.element_size(), n))
    y = torch.empty((d.B, d.Cout, d.Ho, d.Wo), device=x.device, dtype=torch.float32)
    ysc = torch.empty_like(y)
    ws = workspace(n, x.device) if packed is None else None
    _hip.check(_hip.lib().air_conv2d_fwd_s2_pair(
        ctypes.byref(d), dptr(x), dptr(w), dptr(w_sc), dptr(packed, torch.uint8, allow_none=True), dptr(y), dptr(ysc),
        dptr(ws, torch.uint8, allow_none=True), csz(n if packed is None else 0), stream()), "air_conv2d_fwd_s2_pair")
    return y, ysc


def conv2d_wgrad(x, dy, w_shape, stride=1, padding=0, in_scale=None, in_shift=None, relu=False,
                 out=None):
    d = _conv_desc(x.shape, w_shape, stride, padding)
    dw = out if out is not None else torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32)
    ws, n = _conv_ws(d, x.device)
    _hip.check(_hip.lib().air_conv2d_wgrad(
        ctypes.byref(d), dptr(x), dptr(dy), dptr(dw), dptr(in_scale, allow_none=True),
        dptr(in_shift, allow_none=True), ci(1 if relu else 0), dptr(ws, torch.uint8), csz(n),
        stream()), "air_conv2d_wgrad")
    return dw


def _bcs(x):
    B, C = x.shape[0], x.shape[1]
    S = x.numel() // (B * C)
    return B, C, S


def bn_stats(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, stats_in=None):
    """Training-mode batch statistics.  Returns (mean, invstd, scale, shift); updates
    running stats in place when given.  stats_in: the records ``conv2d_fwd(..., stats=True)`` returned for THIS x
    (merged in fp64 instead of reading x)."""
    B, C, S = _bcs(x)
    coef = torch.empty((4, C), device=x.device, dtype=torch.float32)
    lib = _hip.lib()
    n = lib.air_bn_ws_bytes(ci(B), ci(C), ci(S))
    ws = workspace(n, x.device)
    _hip.check(lib.air_bn_stats(
        dptr(x), ci(B), ci(C), ci(S), dptr(stats_in, torch.uint8, allow_none=True), dptr(gamma), dptr(beta), cf(eps),
        cf(momentum), dptr(running_mean, allow_none=True), dptr(running_var, allow_none=True),
        dptr(coef[0]), dptr(coef[1]), dptr(coef[2]), dptr(coef[3]), dptr(ws, torch.uint8), csz(n),
        stream()), "air_bn_stats")
    return coef[0], coef[1], coef[2], coef[3]


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=1e-5):
    C = gamma.numel()
    coef = torch.empty((2, C), device=gamma.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_bn_eval_coeffs(dptr(gamma), dptr(beta), dptr(running_mean),
                                             dptr(running_var), cf(eps), ci(C), dptr(coef[0]),
                                             dptr(coef[1]), stream()), "air_bn_eval_coeffs")
    return coef[0], coef[1]


def bn_apply(x, scale, shift, relu=False, out=None, rowmean=None, y_bf=None):
    """y = x * scale[c] + shift[c] (+ ReLU); rowmean (B, C), 32 <= S <= 1024: also the mean over s of every
    output plane (the SE squeeze taken on the way out)."""
    B, C, S = _bcs(x)
    y = out if out is not None else torch.empty_like(x)
    _hip.check(_hip.lib().air_bn_apply_ex(dptr(x), ci(B), ci(C), ci(S), dptr(scale), dptr(shift),
                                          ci(1 if relu else 0), dptr(y), dptr(rowmean, allow_none=True),
                                          dptr(y_bf, torch.int16, allow_none=True),
                                          ci(y_bf.shape[2] if y_bf is not None else 0), stream()),
               "air_bn_apply_ex")
    return y


def bn_bwd(x, dy, mean, invstd, gamma, beta, relu=False, dx=None, accumulate=False,
           dgamma=None, dbeta=None, relu_in=False, rowbias=None, rowbias_scale=1.0, dbias=None, dy2=None,
           dx_bf16=None, sums_in=None):
    """Backward of y = relu?(batchnorm_train(x)).  Returns (dx, dgamma, dbeta).
    relu_in: x is itself a ReLU output (conv -> ReLU -> BN); dx is then the gradient
    w.r.t. the pre-ReLU tensor.  rowbias (B, C): the incoming gradient is dy + rowbias_scale *
    rowbias[b, c] (broadcast over time).  dbias (C,): receives sum_{b,t} dx, the conv-bias gradient.
    dy (and the optional second gradient dy2, added on the fly) may be channel-slice views.
    dx_bf16: (B, C, Tp) int16 buffer from bf16_rows(): receives dx rounded to bf16, the operand of conv1d_wgrad."""
    B, C, S = _bcs(x)
    if dy.dim() == 3 and (dy2 is not None or not dy.is_contiguous()):
        dyp, dyb = vptr(dy)
        dy2p, dy2b = vptr(dy2) if dy2 is not None else (ctypes.c_void_p(0), 0)
    else:
        dyp, dyb, dy2p, dy2b = dptr(dy), 0, ctypes.c_void_p(0), 0
    if dx is None:
        if accumulate:
            raise _hip.AirError("bn_bwd: accumulate needs an existing dx")
        dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
    if dbeta is None:
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
    lib = _hip.lib()
    n = lib.air_bn_ws_bytes(ci(B), ci(C), ci(S))
    ws = workspace(n, x.device)
    if dx_bf16 is not None and tuple(dx_bf16.shape[:2]) != (B, C):
        raise _hip.AirError("bn_bwd: dx_bf16 must be (B, C, Tp)")
    # sums_in: the records conv2d_dgrad(..., bn=...) took for THIS (x, dy): the pass that re-reads both is skipped
    _hip.check(lib.air_bn_bwd_ex3(dptr(x), dyp, csz(dyb), dy2p, csz(dy2b), dptr(rowbias, allow_none=True),
                                  cf(rowbias_scale), ci(B), ci(C), ci(S), dptr(mean), dptr(invstd), dptr(gamma),
                                  dptr(beta),
                                  ci((1 if relu else 0) | (2 if relu_in else 0)), dptr(dx),
                                  ci(1 if accumulate else 0), dptr(dgamma), dptr(dbeta),
                                  dptr(dbias, allow_none=True), dptr(dx_bf16, torch.int16, allow_none=True),
                                  ci(dx_bf16.shape[2] if dx_bf16 is not None else 0),
                                  dptr(sums_in, torch.uint8, allow_none=True),
                                  dptr(ws, torch.uint8), csz(n), stream()),
               "air_bn_bwd_ex3")
    return dx, dgamma, dbeta


def selfatt_pool_fwd(x, att_w, noise=None):
    """x: (B, C, T).  Returns (out (B, 2C), alpha (B, T))."""
    B, C, T = x.shape
    out = torch.empty((B, 2 * C), device=x.device, dtype=torch.float32)
    alpha = torch.empty((B, T), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_selfatt_pool_fwd(dptr(x), ci(B), ci(C), ci(T), dptr(att_w),
                                               dptr(noise, allow_none=True), dptr(out), dptr(alpha),
                                               stream()), "air_selfatt_pool_fwd")
    return out, alpha


def selfatt_pool_bwd(x, att_w, noise, alpha, out, dout):
    """Returns (dx (B,C,T), datt_partial (B,C))."""
    B, C, T = x.shape
    dx = torch.empty_like(x)
    datt = torch.empty((B, C), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_selfatt_pool_bwd(dptr(x), ci(B), ci(C), ci(T), dptr(att_w),
                                               dptr(noise, allow_none=True), dptr(alpha), dptr(out),
                                               dptr(dout), dptr(dx), dptr(datt), stream()),
               "air_selfatt_pool_bwd")
    return dx, datt


def linear_fwd(x, w, b=None, relu=False):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    fn = _hip.lib().air_linear_relu_fwd if relu else _hip.lib().air_linear_fwd
    _hip.check(fn(dptr(x), dptr(w), dptr(b, allow_none=True), ci(M), ci(K), ci(N), dptr(y),
                  stream()), "air_linear_fwd")
    return y


def linear_bwd(x, w, dy, need_dx=True, dw=None, db=None, need_db=True):
    M, K = x.shape
    N = w.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    if dw is None:
        dw = torch.empty_like(w)
    if db is None and need_db:
        db = torch.empty(N, device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_linear_bwd(dptr(x), dptr(w), dptr(dy), ci(M), ci(K), ci(N),
                                         dptr(dx, allow_none=True), dptr(dw),
                                         dptr(db, allow_none=True), stream()), "air_linear_bwd")
    return dx, dw, db


def ocsoftmax_fwd(x, center, labels, r_real, r_fake, alpha):
    B, D = x.shape
    loss = torch.empty((), device=x.device, dtype=torch.float32)
    neg = torch.empty(B, device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_ocsoftmax_fwd(dptr(x), dptr(center), dptr(labels, torch.int64), ci(B),
                                            ci(D), cf(r_real), cf(r_fake), cf(alpha), dptr(loss),
                                            dptr(neg), stream()), "air_ocsoftmax_fwd")
    return loss, neg


def ocsoftmax_bwd(x, center, labels, r_real, r_fake, alpha, gscale=None, dcenter=None):
    B, D = x.shape
    dx = torch.empty_like(x)
    if dcenter is None:
        dcenter = torch.empty_like(center)
    _hip.check(_hip.lib().air_ocsoftmax_bwd(dptr(x), dptr(center), dptr(labels, torch.int64), ci(B),
                                            ci(D), cf(r_real), cf(r_fake), cf(alpha),
                                            dptr(gscale, allow_none=True), dptr(dx), dptr(dcenter),
                                            stream()), "air_ocsoftmax_bwd")
    return dx, dcenter


def adam_step(p, g, m, v, step, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4,
              grad_scale=1.0):
    _hip.check(_hip.lib().air_adam_step(dptr(p), dptr(g), dptr(m), dptr(v), csz(p.numel()),
                                        ci(step), cf(lr), cf(beta1), cf(beta2), cf(eps),
                                        cf(weight_decay), cf(grad_scale), stream()), "air_adam_step")


def sgd_step(p, g, lr, grad_scale=1.0):
    _hip.check(_hip.lib().air_sgd_step(dptr(p), dptr(g), csz(p.numel()), cf(lr), cf(grad_scale),
                                       stream()), "air_sgd_step")


def add_(y, x):
    _hip.check(_hip.lib().air_add_inplace(dptr(y), dptr(x), csz(y.numel()), stream()),
               "air_add_inplace")
    return y


def sum_rows(x, out=None):
    M, N = x.shape
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_sum_rows(dptr(x), ci(M), ci(N), dptr(out), stream()), "air_sum_rows")
    return out


def randn(shape, device, seed, offset, scale=1.0):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    _hip.check(_hip.lib().air_randn(dptr(out), csz(out.numel()), ctypes.c_uint64(seed),
                                    ctypes.c_uint64(offset), cf(scale), stream()), "air_randn")
    return out


def randn_ctr(shape, device, seed, counter, scale=1.0):
    """randn() with the stream offset in device memory: ``counter`` is a 1-element int64 GPU tensor that the call reads
    and then advances by ceil(numel / 4) on the stream - capturable in a hipGraph, and the same sequence eagerly."""
    if counter.dtype != torch.int64 or not counter.is_cuda or counter.numel() != 1:
        raise _hip.AirError("randn_ctr: counter must be a 1-element int64 GPU tensor")
    out = torch.empty(shape, device=device, dtype=torch.float32)
    _hip.check(_hip.lib().air_randn_ctr(dptr(out), csz(out.numel()), ctypes.c_uint64(seed),
                                        dptr(counter, torch.int64), cf(scale), stream()), "air_randn_ctr")
    return out


# ------------------------------------------------------------------ ECAPA (B, C, T) ops
def vptr(t):
    """(pointer, batch stride) of a (B, C, T) tensor or channel-slice view whose (C, T)
    block is dense."""
    if t.dim() != 3 or not t.is_cuda or t.dtype != torch.float32:
        raise _hip.AirError("expected a (B, C, T) fp32 GPU tensor")
    B, C, T = t.shape
    if t.stride(2) != 1 or (C > 1 and t.stride(1) != T):
        raise _hip.AirError("channel-slice view must keep (C, T) dense")
    return ctypes.c_void_p(t.data_ptr()), (t.stride(0) if B > 1 else C * T)


def _c1d(x, cout, K, dil, pad, y=None):
    B, Cin, T = x.shape
    xp, xb = vptr(x)
    yb = vptr(y)[1] if y is not None else cout * T
    d = AirConv1d(B, Cin, T, cout, K, dil, pad, xb, yb)
    n = _hip.lib().air_conv1d_ws_bytes(ctypes.byref(d))
    if n == 0:
        raise _hip.AirError("conv1d: unsupported configuration K=%d dil=%d pad=%d" % (K, dil, pad))
    return d, workspace(n, x.device), n


def _c1d_bf16(d, device, which):
    """Workspace for the bf16 kernels (pointwise layers; dilated K = 3 forward / dgrad), or None when the
    layer is not theirs (K = 5, ragged channel counts, K = 3 wgrad): the caller then runs the fp32 kernels."""
    lib = _hip.lib()
    if not lib.air_conv1d_bf16_supported(ctypes.byref(d), ci(which)):
        return None, 0
    n = lib.air_conv1d_bf16_ws_bytes(ctypes.byref(d))
    return workspace(n, device), n


def conv1d_tap_pack(weights, transpose, out=None):
    """bf16 operand blocks of a list of equally shaped K = 3 conv weights, packed in ONE launch when they sit at a
    regular stride in memory (the Res2 branch convs inside the parameter arena) and one launch per weight otherwise
    (parameters re-assigned outside the arena).  Returns an (n, 3*Cout*Cin) int16 tensor whose rows are the w_packed
    arguments of conv1d_fwd / conv1d_dgrad / ops_h.conv_tap; None only for shapes the tap kernels do not take."""
    w0 = weights[0]
    n = len(weights)
    Cout, Cin, K = w0.shape
    if K != 3 or any(w.shape != w0.shape or w.dtype != torch.float32 or not w.is_cuda for w in weights):
        return None
    weights = [w if w.is_contiguous() else w.contiguous() for w in weights]
    w0 = weights[0]
    stride = (weights[1].data_ptr() - w0.data_ptr()) // 4 if n > 1 else 0
    regular = n == 1 or (stride > 0 and all(w.data_ptr() == w0.data_ptr() + 4 * stride * i for i, w in enumerate(weights)))
    per = Cout * Cin * 3
    if out is None:
        out = torch.empty((n, per), device=w0.device, dtype=torch.int16)
    tr = ci(1 if transpose else 0)
    if regular:
        _hip.check(_hip.lib().air_conv1d_tap_pack_bf16(dptr(w0), csz(stride), ci(n), ci(Cout), ci(Cin), tr,
                                                       dptr(out, torch.int16), stream()), "air_conv1d_tap_pack_bf16")
    else:
        for i, w in enumerate(weights):
            _hip.check(_hip.lib().air_conv1d_tap_pack_bf16(dptr(w), csz(0), ci(1), ci(Cout), ci(Cin), tr,
                                                           dptr(out[i], torch.int16), stream()),
                       "air_conv1d_tap_pack_bf16")
    return out


def conv1d_fwd(x, w, bias=None, bias_bc=None, relu=False, dil=1, pad=0, out=None, bf16=False, y_bf=None,
               w_packed=None):
    """y = relu?(conv1d(x, w) + bias + bias_bc[b]); x / out may be channel-slice views.
    bf16: pointwise layers run on the bf16 matrix cores (operands rounded, fp32 accumulate); y_bf (bf16_rows
    buffer, bf16 path only) also receives y rounded to bf16 - the X operand of a later conv1d_wgrad."""
    Cout, Cin, K = w.shape
    B, _, T = x.shape
    y = out if out is not None else torch.empty((B, Cout, T), device=x.device, dtype=torch.float32)
    d, ws, n = _c1d(x, Cout, K, dil, pad, y)
    if bf16:
        wsb, nb = _c1d_bf16(d, x.device, 0)
        if wsb is not None:
            _hip.check(_hip.lib().air_conv1d_fwd_bf16_ex(
                ctypes.byref(d), vptr(x)[0], dptr(w), dptr(w_packed, torch.int16, allow_none=True),
                dptr(bias, allow_none=True),
                dptr(bias_bc, allow_none=True), ci(1 if relu else 0), vptr(y)[0],
                dptr(y_bf, torch.int16, allow_none=True),
                dptr(wsb, torch.uint8), csz(nb), stream()), "air_conv1d_fwd_bf16_ex")
            return y
    if y_bf is not None:
        raise _hip.AirError("conv1d_fwd: y_bf needs the bf16 path")
    _hip.check(_hip.lib().air_conv1d_fwd(ctypes.byref(d), vptr(x)[0], dptr(w), dptr(bias, allow_none=True),
                                         dptr(bias_bc, allow_none=True), ci(1 if relu else 0), vptr(y)[0],
                                         dptr(ws, torch.uint8), csz(n), stream()), "air_conv1d_fwd")
    return y


def conv1d_pointwise_kmajor(xb, w, T, dgrad=False, bias=None, bias_bc=None, relu=False, accumulate=None, out=None,
                            y_bf=None, accumulate2=None):
    """K = 1 forward (y = W . x) or data gradient (dx = W^T . dy + accumulate) straight from the bf16 copy ``xb``
    ((B, C, Tp) int16, or a channel slice of one) of the input operand.  Returns None when the layer does not fit the
    K-major GEMM (the caller falls back to conv1d_fwd / conv1d_dgrad on the fp32 tensor)."""
    Cout, Cin, K = w.shape
    B = xb.shape[0]
    Cy = Cin if dgrad else Cout
    if K != 1 or xb.shape[1] != (Cout if dgrad else Cin):
        return None
    y = out if out is not None else torch.empty((B, Cy, T), device=xb.device, dtype=torch.float32)
    yp, yb = vptr(y)
    d = AirConv1d(B, Cin, T, Cout, 1, 1, 0, yb if dgrad else 0, 0 if dgrad else yb)
    wsb, nb = _c1d_bf16(d, xb.device, 1 if dgrad else 0)
    if wsb is None:
        return None
    xp, xbs_ = _bf_view(xb)
    acc, accb = vptr(accumulate) if accumulate is not None else (ctypes.c_void_p(0), 0)
    acc2, acc2b = vptr(accumulate2) if accumulate2 is not None else (ctypes.c_void_p(0), 0)
    rc = _hip.lib().air_conv1d_pointwise_bf16_kmajor(
        ctypes.byref(d), xp, csz(xbs_), dptr(w), ci(1 if dgrad else 0), dptr(bias, allow_none=True),
        dptr(bias_bc, allow_none=True), ci(1 if relu else 0), acc, csz(accb), acc2, csz(acc2b), yp,
        dptr(y_bf, torch.int16, allow_none=True),
        dptr(wsb, torch.uint8), csz(nb), stream())
    if rc == -2:  # AIR_EUNSUPPORTED: shape outside the K-major kernel
        return None
    _hip.check(rc, "air_conv1d_pointwise_bf16_kmajor")
    return y


def conv1d_dgrad(dy, w, dil=1, pad=0, accumulate=None, out=None, bf16=False, accumulate2=None, w_packed=None):
    """dx = conv1d_transpose(dy, w) (+ accumulate, addressed like out).  bf16 pointwise path: accumulate may be a
    channel-slice view and a second operand accumulate2 (view or dense) is added in the same epilogue."""
    Cout, Cin, K = w.shape
    B, _, T = dy.shape
    dx = out if out is not None else torch.empty((B, Cin, T), device=dy.device, dtype=torch.float32)
    xp, xb = vptr(dx)
    yp, yb = vptr(dy)
    d = AirConv1d(B, Cin, T, Cout, K, dil, pad, xb, yb)
    acc, accb = vptr(accumulate) if accumulate is not None else (ctypes.c_void_p(0), 0)
    if bf16:
        wsb, nb = _c1d_bf16(d, dy.device, 1)
        if wsb is not None:
            acc2, acc2b = vptr(accumulate2) if accumulate2 is not None else (ctypes.c_void_p(0), 0)
            _hip.check(_hip.lib().air_conv1d_dgrad_bf16_ex(ctypes.byref(d), yp, dptr(w),
                                                           dptr(w_packed, torch.int16, allow_none=True), xp, acc,
                                                           csz(accb), acc2,
                                                           csz(acc2b), dptr(wsb, torch.uint8), csz(nb), stream()),
                       "air_conv1d_dgrad_bf16_ex")
            return dx
    if accumulate2 is not None or (accumulate is not None and accb != xb):
        raise _hip.AirError("conv1d_dgrad: a second / strided accumulate operand needs the bf16 pointwise path")
    n = _hip.lib().air_conv1d_ws_bytes(ctypes.byref(d))
    ws = workspace(n, dy.device)
    _hip.check(_hip.lib().air_conv1d_dgrad(ctypes.byref(d), yp, dptr(w), xp, acc, dptr(ws, torch.uint8),
                                           csz(n), stream()), "air_conv1d_dgrad")
    return dx


_BF_ROWS = {}


def bf16_rows(tag, B, C, T, device):
    """Zero-padded (B, C, Tp) int16 buffer for a bf16 operand copy (Tp = air_conv1d_bf16_tp(T)).  With a tag: one
    persistent buffer per (tag, shape), for copies made and consumed inside one backward pass - writers only
    touch frames < T, so the padding stays zero for as long as the buffer lives.  tag None: fresh memory."""
    Tp = int(_hip.lib().air_conv1d_bf16_tp(ci(T)))
    if tag is None:  # a copy that outlives the call sequence (saved for backward): its own memory, padding zeroed
        buf = torch.empty((B, C, Tp), device=device, dtype=torch.int16)
        if Tp > T:
            buf[:, :, T:].zero_()
        return buf
    key = (tag, B, C, Tp, device.type, device.index)
    buf = _BF_ROWS.get(key)
    if buf is None:
        buf = _BF_ROWS[key] = torch.zeros((B, C, Tp), device=device, dtype=torch.int16)
    return buf


def conv1d_cvt_bf16(x, out):
    """out (B, C, Tp) int16 <- bf16(x) for a (B, C, T) fp32 tensor or channel-slice view."""
    B, C, T = x.shape
    xp, xb = vptr(x)
    _hip.check(_hip.lib().air_conv1d_cvt_bf16(xp, csz(xb), ci(B), ci(C), ci(T), dptr(out, torch.int16), stream()),
               "air_conv1d_cvt_bf16")
    return out


def _bf_view(t):
    """(pointer, batch stride in elements) of a (B, C, Tp) int16 copy or a channel-slice view of one."""
    if t is None:
        return ctypes.c_void_p(0), 0
    if t.dtype != torch.int16 or not t.is_cuda or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
        raise _hip.AirError("bf16 operand copy must be a (B, C, Tp) int16 GPU tensor (or a channel slice of one)")
    return ctypes.c_void_p(t.data_ptr()), t.stride(0)


def conv1d_wgrad(x, dy, w_shape, dil=1, pad=0, out=None, bf16=False, x_bf=None, dy_bf=None):
    """x_bf / dy_bf (bf16 path only): operand copies the caller already holds (bf16_rows buffers filled by
    bn_bwd(dx_bf16=...) or conv1d_cvt_bf16), skipping the conversion pass inside."""
    Cout, Cin, K = w_shape
    B, _, T = x.shape
    dw = out if out is not None else torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32)
    xp, xb = vptr(x)
    yp, yb = vptr(dy)
    d = AirConv1d(B, Cin, T, Cout, K, dil, pad, xb, yb)
    if bf16:
        wsb, nb = _c1d_bf16(d, x.device, 2)
        if wsb is not None:
            xfp, xfb = _bf_view(x_bf)
            yfp, yfb = _bf_view(dy_bf)
            _hip.check(_hip.lib().air_conv1d_wgrad_bf16_pre(ctypes.byref(d), xp, yp, xfp, csz(xfb), yfp, csz(yfb),
                                                            dptr(dw), dptr(wsb, torch.uint8), csz(nb), stream()),
                       "air_conv1d_wgrad_bf16_pre")
            return dw
    n = _hip.lib().air_conv1d_ws_bytes(ctypes.byref(d))
    ws = workspace(n, x.device)
    _hip.check(_hip.lib().air_conv1d_wgrad(ctypes.byref(d), xp, yp, dptr(dw), dptr(ws, torch.uint8),
                                           csz(n), stream()), "air_conv1d_wgrad")
    return dw


def add_strided(out, a, b=None, out_bf=None):
    """out = a (+ b) over (B, C, T) tensors / channel-slice views; out_bf: (B, C, Tp) int16 copy (or a channel slice
    of one) that also receives the result as bf16."""
    B, C, T = out.shape
    op, ob = vptr(out)
    ap, ab = vptr(a)
    bp, bb = vptr(b) if b is not None else (ctypes.c_void_p(0), 0)
    fp, fb = _bf_view(out_bf)
    _hip.check(_hip.lib().air_add_strided_ex(op, csz(ob), ap, csz(ab), bp, csz(bb), ci(B), ci(C), ci(T), fp, csz(fb),
                                             ci(out_bf.shape[2] if out_bf is not None else 0), stream()),
               "air_add_strided_ex")
    return out


def res2_bn_apply(x, scale, shift, y1, add=None, y2=None, y1_bf=None):
    """y1 (channel-slice view) = x*scale + shift; y2 (dense) = that + add (channel-slice view); y1_bf: channel slice
    of a (B, C, Tp) int16 copy that also receives y1 as bf16."""
    B, C, T = x.shape
    y1p, y1b = vptr(y1)
    ap, ab = vptr(add) if add is not None else (ctypes.c_void_p(0), 0)
    fp, fb = _bf_view(y1_bf)
    _hip.check(_hip.lib().air_res2_bn_apply_ex(dptr(x), ci(B), ci(C), ci(T), dptr(scale), dptr(shift), y1p, csz(y1b),
                                               ap, csz(ab), dptr(y2, allow_none=True), fp, csz(fb),
                                               ci(y1_bf.shape[2] if y1_bf is not None else 0), stream()),
               "air_res2_bn_apply_ex")
    return y2


def channel_sum(x, out=None):
    B, C, T = x.shape
    xp, xb = vptr(x)
    if out is None:
        out = torch.empty(C, device=x.device, dtype=torch.float32)
    n = _hip.lib().air_channel_sum_ws_bytes(ci(B), ci(C))
    ws = workspace(n, x.device)
    _hip.check(_hip.lib().air_channel_sum(xp, ci(B), ci(C), ci(T), csz(xb), dptr(out), dptr(ws, torch.uint8),
                                          csz(n), stream()), "air_channel_sum")
    return out


def row_stats(x, want_std=True, clamp_min=1e-4, mean_out=None, std_out=None):
    B, C, T = x.shape
    mean = mean_out if mean_out is not None else torch.empty((B, C), device=x.device, dtype=torch.float32)
    std = None
    if want_std:
        std = std_out if std_out is not None else torch.empty((B, C), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_row_stats(dptr(x), ci(B), ci(C), ci(T), dptr(mean), dptr(std, allow_none=True),
                                        cf(clamp_min), stream()), "air_row_stats")
    return mean, std


def row_stats_bwd(x, mean, std, dmean, dstd, dx, accumulate=True, clamp_min=1e-4, relu_mask=False, rowsum=None,
                  dx_bf16=None):
    """dx (+)= gradient of the per-row mean / std statistics; relu_mask zeroes the result where x == 0;
    rowsum (B, C) receives the time sums of the result rows; dx_bf16 (bf16_rows buffer) its bf16 copy."""
    B, C, T = x.shape
    _hip.check(_hip.lib().air_row_stats_bwd_ex(dptr(x), ci(B), ci(C), ci(T), dptr(mean), dptr(std, allow_none=True),
                                               dptr(dmean, allow_none=True), dptr(dstd, allow_none=True),
                                               cf(clamp_min), dptr(dx), ci(1 if accumulate else 0),
                                               ci(1 if relu_mask else 0), dptr(rowsum, allow_none=True),
                                               dptr(dx_bf16, torch.int16, allow_none=True),
                                               ci(dx_bf16.shape[2] if dx_bf16 is not None else 0), stream()),
               "air_row_stats_bwd_ex")
    return dx


def row_sum(x):
    B, C, T = x.shape
    out = torch.empty((B, C), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_row_sum(dptr(x), ci(B), ci(C), ci(T), dptr(out), stream()), "air_row_sum")
    return out


def relu_mask_(dx, y):
    _hip.check(_hip.lib().air_relu_mask(dptr(dx), dptr(y), csz(dx.numel()), stream()), "air_relu_mask")
    return dx


def se_scale_fwd(x, z, res, out, out_bf=None):
    """out = x * sigmoid(z[b][c]) + res; out_bf: channel slice of a (B, C', Tp) int16 copy that also receives it as bf16."""
    B, C, T = x.shape
    rp, rb = vptr(res)
    op, ob = vptr(out)
    fp, fb = _bf_view(out_bf)
    _hip.check(_hip.lib().air_se_scale_fwd_ex(dptr(x), dptr(z), rp, csz(rb), ci(B), ci(C), ci(T), op, csz(ob), fp,
                                              csz(fb), ci(out_bf.shape[2] if out_bf is not None else 0), stream()),
               "air_se_scale_fwd_ex")
    return out


def se_scale_bwd(x, z, dout):
    B, C, T = x.shape
    dp, db = vptr(dout)
    dx = torch.empty_like(x)
    dz = torch.empty_like(z)
    _hip.check(_hip.lib().air_se_scale_bwd(dptr(x), dptr(z), dp, csz(db), ci(B), ci(C), ci(T), dptr(dx),
                                           dptr(dz), stream()), "air_se_scale_bwd")
    return dx, dz


def asp_fwd(x, logits):
    """Overwrites ``logits`` with the softmax weights; returns (B, 2C) = [mu | sg]."""
    B, C, T = x.shape
    out = torch.empty((B, 2 * C), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_asp_fwd(dptr(x), dptr(logits), ci(B), ci(C), ci(T), dptr(out), stream()),
               "air_asp_fwd")
    return out


def asp_bwd(x, w, out, dout, dx, accumulate=False, rowsum=None, dlogits_bf16=None):
    """Overwrites ``w`` with d(logits); writes / accumulates dx; rowsum (B, C) receives sum_t d(logits);
    dlogits_bf16 (bf16_rows buffer) the bf16 copy of d(logits)."""
    B, C, T = x.shape
    _hip.check(_hip.lib().air_asp_bwd_ex(dptr(x), dptr(w), ci(B), ci(C), ci(T), dptr(out), dptr(dout), dptr(dx),
                                         ci(1 if accumulate else 0), dptr(rowsum, allow_none=True),
                                         dptr(dlogits_bf16, torch.int16, allow_none=True),
                                         ci(dlogits_bf16.shape[2] if dlogits_bf16 is not None else 0), stream()),
               "air_asp_bwd_ex")
    return dx


def softmax_rows(logits):
    """softmax over dim 1 of (B, C) logits (generate_score.py:102) on the softmax / cross-entropy kernel."""
    B, C = logits.shape
    logits = logits.float().contiguous()
    probs = torch.empty_like(logits)
    labels = torch.zeros(B, dtype=torch.int64, device=logits.device)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    correct = torch.empty((), device=logits.device, dtype=torch.int32)
    _hip.check(_hip.lib().air_softmax_ce_fwd(_hip.dptr(logits), _hip.dptr(labels, torch.int64), _hip.ci(B), _hip.ci(C),
                                             _hip.dptr(probs), _hip.dptr(loss), _hip.dptr(correct, torch.int32),
                                             _hip.stream()), "air_softmax_ce_fwd")
    return probs


# BatchNorm's num_batches_tracked counters: one multi-tensor add per forward instead of one 4 us ATen launch per
# layer (34 per ECAPA step).  bn_tick() queues a counter, bn_flush() adds 1 to everything queued since the last flush.
_TICKS = []


def bn_tick(counter):
    _TICKS.append(counter)


def bn_flush():
    if _TICKS:
        torch._foreach_add_(_TICKS, 1)
        del _TICKS[:]


class ClockProbe:
    """Core clock of the GPU WHILE other kernels run (air_debug_clock_probe): one wave on a side stream samples
    {wall clock, core-clock counter} every ``interval_us``; ``mhz()`` after the measured work has been synchronised.
    Measurement instrumentation (bench.py "core_clock"), not part of the training path."""

    def __init__(self, device, n_samples=2000, interval_us=50.0):
        self.n = int(n_samples)
        self.buf = torch.zeros(2 * self.n, dtype=torch.int64, device=device)
        self.side = torch.cuda.Stream(device=device)
        self.interval_us = float(interval_us)

    def start(self):
        self.buf.zero_()
        torch.cuda.current_stream(self.buf.device).synchronize()
        _hip.check(_hip.lib().air_debug_clock_probe(ctypes.c_void_p(self.buf.data_ptr()), ci(self.n),
                                                    ctypes.c_double(self.interval_us),
                                                    ctypes.c_void_p(self.side.cuda_stream)), "air_debug_clock_probe")

    def samples(self):
        """(t_us, mhz) per sampling interval, after the probe has finished."""
        self.side.synchronize()
        raw = self.buf.cpu().numpy().reshape(self.n, 2)
        ok = raw[:, 0] > 0
        raw = raw[ok]
        dw = (raw[1:, 0] - raw[:-1, 0]).astype("float64")
        dc = (raw[1:, 1] - raw[:-1, 1]).astype("float64")
        keep = dw > 0
        t = (raw[1:, 0] - raw[0, 0])[keep] / 100.0
        return t, dc[keep] / dw[keep] * 100.0

    def mhz(self, t_lo_us=None, t_hi_us=None):
        """Median / min / max core clock over the samples whose time since the probe's start lies in [t_lo, t_hi]."""
        import numpy as np
        t, f = self.samples()
        if t_lo_us is not None:
            sel = (t >= t_lo_us) & (t <= t_hi_us)
            f = f[sel]
        if f.size == 0:
            return None
        return {"median": round(float(np.median(f)), 1), "min": round(float(f.min()), 1), "max": round(float(f.max()), 1),
                "samples": int(f.size), "interval_us": self.interval_us}

