"""Tensor-level wrappers over the C-ABI (include/air_hip.h).

Each function takes contiguous fp32 GPU tensors, allocates outputs with torch
(device memory is plumbing) and launches the HIP kernels on the current
stream.  Nothing here computes on the CPU or through ATen.
"""
import ctypes

import torch

from . import _hip
from ._hip import AirConv2d, ci, cf, csz, dptr, stream

_WS = {}


def workspace(nbytes, device):
    """One growing scratch buffer per device (ops run back to back on one stream)."""
    key = (device.type, device.index)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
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


def conv2d_fwd(x, w, stride=1, padding=0, in_scale=None, in_shift=None, relu=False, residual=None):
    """y = conv2d(act(x), w) (+ residual); act = optional per-channel affine + ReLU."""
    d = _conv_desc(x.shape, w.shape, stride, padding)
    y = torch.empty((d.B, d.Cout, d.Ho, d.Wo), device=x.device, dtype=torch.float32)
    ws, n = _conv_ws(d, x.device)
    _hip.check(_hip.lib().air_conv2d_fwd(
        ctypes.byref(d), dptr(x), dptr(w), dptr(y), dptr(in_scale, allow_none=True),
        dptr(in_shift, allow_none=True), ci(1 if relu else 0), dptr(residual, allow_none=True),
        ctypes.c_void_p(0), dptr(ws, torch.uint8), csz(n), stream()), "air_conv2d_fwd")
    return y


def conv2d_dgrad(dy, w, x_shape, stride=1, padding=0, accumulate=None, out=None):
    d = _conv_desc(x_shape, w.shape, stride, padding)
    dx = out if out is not None else torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    ws, n = _conv_ws(d, dy.device)
    _hip.check(_hip.lib().air_conv2d_dgrad(
        ctypes.byref(d), dptr(dy), dptr(w), dptr(dx), dptr(accumulate, allow_none=True),
        dptr(ws, torch.uint8), csz(n), stream()), "air_conv2d_dgrad")
    return dx


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


def bn_stats(x, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """Training-mode batch statistics.  Returns (mean, invstd, scale, shift); updates
    running stats in place when given."""
    B, C, S = _bcs(x)
    coef = torch.empty((4, C), device=x.device, dtype=torch.float32)
    lib = _hip.lib()
    n = lib.air_bn_ws_bytes(ci(B), ci(C), ci(S))
    ws = workspace(n, x.device)
    _hip.check(lib.air_bn_stats(
        dptr(x), ci(B), ci(C), ci(S), ctypes.c_void_p(0), dptr(gamma), dptr(beta), cf(eps),
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


def bn_apply(x, scale, shift, relu=False, out=None):
    B, C, S = _bcs(x)
    y = out if out is not None else torch.empty_like(x)
    _hip.check(_hip.lib().air_bn_apply(dptr(x), ci(B), ci(C), ci(S), dptr(scale), dptr(shift),
                                       ci(1 if relu else 0), dptr(y), stream()), "air_bn_apply")
    return y


def bn_bwd(x, dy, mean, invstd, gamma, beta, relu=False, dx=None, accumulate=False,
           dgamma=None, dbeta=None):
    """Backward of y = relu?(batchnorm_train(x)).  Returns (dx, dgamma, dbeta)."""
    B, C, S = _bcs(x)
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
    _hip.check(lib.air_bn_bwd(dptr(x), dptr(dy), ci(B), ci(C), ci(S), dptr(mean), dptr(invstd),
                              dptr(gamma), dptr(beta), ci(1 if relu else 0), dptr(dx),
                              ci(1 if accumulate else 0), dptr(dgamma), dptr(dbeta),
                              dptr(ws, torch.uint8), csz(n), stream()), "air_bn_bwd")
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


def linear_fwd(x, w, b=None):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_linear_fwd(dptr(x), dptr(w), dptr(b, allow_none=True), ci(M), ci(K),
                                         ci(N), dptr(y), stream()), "air_linear_fwd")
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
