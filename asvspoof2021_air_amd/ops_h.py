"""Tensor-level wrappers over the bf16-RESIDENT entry points of the C-ABI (include/air_hip.h, ``air_h_*``).

A resident activation is a ``torch.int16`` tensor of shape (B, C, Tp) holding bf16 bits, Tp = ``tp(T)`` frames per
row with the frames T .. Tp - 1 zero, or a channel-slice view of one (batch stride = the wide tensor's).  ``T`` (the
logical frame count) travels as an argument.  Nothing here computes on the CPU or through ATen.
"""
import ctypes
import functools

import torch

from . import _hip, ops
from ._hip import ci, cf, csz, dptr, stream


def tp(T):
    return int(_hip.lib().air_h_tp(ci(T)))


def max_tp():
    """Longest row (frames, padded) the row kernels of csrc/ecapa_bf16.hip accept (h_shape_ok: 256 lanes x HMAXV)."""
    return 2048


def rows(B, C, T, device, zero=False):
    """Fresh (B, C, Tp) buffer.  Writers fill every frame of every row they touch (zeros behind T)."""
    f = torch.zeros if zero else torch.empty
    return f((B, C, tp(T)), device=device, dtype=torch.int16)


def hv(t, allow_none=False):
    """(pointer, batch stride in elements) of a resident tensor or channel-slice view."""
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0), 0
        raise _hip.AirError("null resident tensor")
    if t.dtype != torch.int16 or not t.is_cuda or t.dim() != 3 or t.stride(2) != 1 or (t.shape[1] > 1 and t.stride(1) != t.shape[2]):
        raise _hip.AirError("resident activation must be a (B, C, Tp) int16 GPU tensor or a channel slice of one")
    return ctypes.c_void_p(t.data_ptr()), (t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2])


def from_f32(x, out=None):
    """(B, C, T) fp32 (dense, or a channel-slice view with a batch stride) -> resident rows, rounded to nearest even:
    ``air_h_unfold`` with a one-tap window (row c = x[c][t], zeros behind T) - one kernel for every T."""
    B, C, T = x.shape
    if x.dtype != torch.float32 or not x.is_cuda or x.stride(2) != 1 or (C > 1 and x.stride(1) != T):
        raise _hip.AirError("from_f32: (B, C, T) fp32 GPU tensor with dense rows expected")
    if out is None:
        out = rows(B, C, T, x.device)
    q, qs = hv(out)
    xs = x.stride(0) if B > 1 else C * T
    _hip.check(_hip.lib().air_h_unfold(dptr(x), csz(xs), ci(B), ci(C), ci(T), ci(out.shape[2]), ci(1), ci(1), ci(0),
                                       ci(C), q, csz(qs), stream()), "air_h_unfold")
    return out


def to_f32(x, T):
    B, C, Tp = x.shape
    y = torch.empty((B, C, T), device=x.device, dtype=torch.float32)
    p, bs = hv(x)
    _hip.check(_hip.lib().air_h_to_f32(p, csz(bs), ci(B), ci(C), ci(T), ci(Tp), dptr(y), stream()), "air_h_to_f32")
    return y


def unfold(x, K, dil, pad, rows_out):
    """Dense fp32 (B, Cin, T) -> resident (B, rows_out, Tp): row ci * K + k = x[ci][t + k * dil - pad] (zeros outside)."""
    B, Cin, T = x.shape
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise _hip.AirError("unfold: dense fp32 input expected")
    out = rows(B, rows_out, T, x.device)
    _hip.check(_hip.lib().air_h_unfold(dptr(x), csz(0), ci(B), ci(Cin), ci(T), ci(out.shape[2]), ci(K), ci(dil), ci(pad),
                                       ci(rows_out), dptr(out, torch.int16), csz(0), stream()), "air_h_unfold")
    return out


def copy(x, out):
    B, C, Tp = x.shape
    p, bs = hv(x)
    q, qs = hv(out)
    _hip.check(_hip.lib().air_h_copy(p, csz(bs), ci(B), ci(C), ci(Tp), q, csz(qs), stream()), "air_h_copy")
    return out


def conv_pointwise(x, w, T, dgrad=False, bias=None, bias_bc=None, relu=False, acc=None, acc2=None, out=None, stats=False):
    """K = 1 conv on resident rows.  w: the layer's (Cout, Cin, 1) fp32 weight.  Forward: x (B, Cin, Tp) ->
    (B, Cout, Tp); dgrad: x = dy (B, Cout, Tp) -> (B, Cin, Tp) (+ acc + acc2, resident rows / slices).
    stats=True (forward): returns (y, records) - BatchNorm statistics of the stored y for ``bn_stats(stats_in=...)``."""
    Cout, Cin = w.shape[0], w.shape[1]
    B, K, Tp = x.shape
    M = Cin if dgrad else Cout
    if K != (Cout if dgrad else Cin):
        raise _hip.AirError("conv_pointwise: operand has %d channels" % K)
    if out is None:
        out = torch.empty((B, M, Tp), device=x.device, dtype=torch.int16)
    lib = _hip.lib()
    n = _pointwise_ws_bytes(Cout, Cin)
    ws = ops.workspace(n, x.device)
    xp, xb = hv(x)
    ap, ab = hv(acc, True)
    a2p, a2b = hv(acc2, True)
    yp, yb = hv(out)
    rec = None
    if stats:
        if dgrad:
            raise _hip.AirError("conv_pointwise: statistics records are a forward-launch feature")
        rec = torch.empty(int(lib.air_h_conv1d_pointwise_stats_bytes(ci(B), ci(M), ci(Tp))), dtype=torch.uint8, device=x.device)
    _hip.check(lib.air_h_conv1d_pointwise_ex(ci(B), ci(Cin), ci(Cout), ci(T), ci(Tp), xp, csz(xb), dptr(w),
                                             ci(1 if dgrad else 0), dptr(bias, allow_none=True),
                                             dptr(bias_bc, allow_none=True), ci(1 if relu else 0), ap, csz(ab), a2p, csz(a2b),
                                             yp, csz(yb), dptr(rec, torch.uint8, allow_none=True), dptr(ws, torch.uint8),
                                             csz(n), stream()), "air_h_conv1d_pointwise_ex")
    return (out, rec) if stats else out


def conv_wgrad(x, dy, T, out):
    """out (Cout, Cin, 1) fp32 = sum_{b,t} dy x over resident operands."""
    B, Cin, Tp = x.shape
    Cout = dy.shape[1]
    lib = _hip.lib()
    n = _wgrad_ws_bytes(B, Cin, T, Cout)
    ws = ops.workspace(n, x.device)
    xp, xb = hv(x)
    yp, yb = hv(dy)
    _hip.check(lib.air_h_conv1d_wgrad(ci(B), ci(Cin), ci(Cout), ci(T), ci(Tp), xp, csz(xb), yp, csz(yb), dptr(out),
                                      dptr(ws, torch.uint8), csz(n), stream()), "air_h_conv1d_wgrad")
    return out


class AirTapPrologue(ctypes.Structure):
    """include/air_hip.h: the elementwise pass that produces a Res2 conv's operand, folded into its staging."""
    _fields_ = [("kind", ctypes.c_int),
                ("g0", ctypes.c_void_p), ("g0_bs", ctypes.c_size_t), ("g1", ctypes.c_void_p), ("g1_bs", ctypes.c_size_t),
                ("pa", ctypes.c_void_p), ("pb", ctypes.c_void_p), ("pc", ctypes.c_void_p), ("pd", ctypes.c_void_p),
                ("pe", ctypes.c_void_p),
                ("side0", ctypes.c_void_p), ("side0_bs", ctypes.c_size_t), ("side1", ctypes.c_void_p),
                ("side1_bs", ctypes.c_size_t)]


def tap_pro_ok(Cin, Cout):
    return bool(_hip.lib().air_h_conv1d_tap_pro_ok(ci(Cin), ci(Cout)))


def _f32p(t):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise _hip.AirError("prologue coefficients must be contiguous fp32 GPU tensors")
    return t.data_ptr()


def res2_prologue(scale, shift, add, y1, t_out):
    """Forward: the conv's input is r of the PREVIOUS branch; operand = bf16(bf16(r * scale + shift) + add); y1 (the
    previous branch's slice of the concat) and t_out (the operand, kept for the weight gradient) are written."""
    g0, g0b = hv(add)
    s0, s0b = hv(t_out)
    s1, s1b = hv(y1)
    return AirTapPrologue(1, g0.value, g0b, None, 0, _f32p(scale), _f32p(shift), None, None, None, s0.value, s0b, s1.value, s1b)


def bn_bwd_prologue(dy, dy2, mean, invstd, gamma, dgamma, dbeta, dc_out):
    """Data gradient: the conv's input is r of THIS branch (its BatchNorm's input); operand = the BatchNorm-backward
    apply of (dy + dy2) with the ReLU mask; dc_out receives it (the weight gradient's dy)."""
    g0, g0b = hv(dy)
    g1, g1b = hv(dy2, True)
    s0, s0b = hv(dc_out)
    return AirTapPrologue(2, g0.value, g0b, g1.value, g1b, _f32p(mean), _f32p(invstd), _f32p(gamma), _f32p(dbeta),
                          _f32p(dgamma), s0.value, s0b, None, 0)


def conv_tap(x, w_packed, T, dil, Cout, Cin, dgrad=False, bias=None, relu=False, out=None, stats=False, bn=None, pro=None):
    """stats=True: returns (y, records) - the BatchNorm statistics of the stored y from the epilogue, for
    ``bn_stats(y, ..., stats_in=records)``.  bn = (bn_x, bn_dy, mean, invstd) on a data-gradient launch: returns
    (y, sums) - the backward sums of the BatchNorm whose output gradient is bn_dy + y, for ``bn_bwd(..., sums_in=sums)``.
    pro (round 6): ``res2_prologue`` / ``bn_bwd_prologue`` - x is then the RAW tensor the operand is computed from."""
    B, K, Tp = x.shape
    M = Cin if dgrad else Cout
    if out is None:
        out = torch.empty((B, M, Tp), device=x.device, dtype=torch.int16)
    xp, xb = hv(x)
    yp, yb = hv(out)
    lib = _hip.lib()
    rec = sums = None
    if stats:
        rec = torch.empty(int(lib.air_h_conv1d_tap_stats_bytes(ci(B), ci(M), ci(Tp))), dtype=torch.uint8, device=x.device)
    bxp, bxb, bdp, bdb, mean, invstd = None, 0, None, 0, None, None
    if bn is not None:
        bn_x, bn_dy, mean, invstd = bn
        sums = torch.empty(int(lib.air_h_conv1d_tap_bwd_sums_bytes(ci(B), ci(M), ci(Tp))), dtype=torch.uint8, device=x.device)
        bxp, bxb = hv(bn_x)
        bdp, bdb = hv(bn_dy)
    if pro is not None:
        _hip.check(lib.air_h_conv1d_tap_pro(ci(B), ci(Cin), ci(Cout), ci(T), ci(Tp), ci(dil), xp, csz(xb),
                                            dptr(w_packed, torch.int16), ci(1 if dgrad else 0), dptr(bias, allow_none=True),
                                            ci(1 if relu else 0), yp, csz(yb), dptr(rec, torch.uint8, allow_none=True),
                                            bxp, csz(bxb), bdp, csz(bdb), dptr(mean, allow_none=True),
                                            dptr(invstd, allow_none=True), dptr(sums, torch.uint8, allow_none=True),
                                            ctypes.byref(pro), stream()), "air_h_conv1d_tap_pro")
    else:
        _hip.check(lib.air_h_conv1d_tap_ex2(ci(B), ci(Cin), ci(Cout), ci(T), ci(Tp), ci(dil), xp, csz(xb),
                                            dptr(w_packed, torch.int16), ci(1 if dgrad else 0), dptr(bias, allow_none=True),
                                            ci(1 if relu else 0), yp, csz(yb), dptr(rec, torch.uint8, allow_none=True),
                                            bxp, csz(bxb), bdp, csz(bdb), dptr(mean, allow_none=True),
                                            dptr(invstd, allow_none=True), dptr(sums, torch.uint8, allow_none=True), stream()),
                   "air_h_conv1d_tap_ex2")
    if bn is not None:
        return out, sums
    return (out, rec) if stats else out


def conv_tap_wgrad(xs, dys, T, dil, outs):
    """outs[i] (W, W, 3) fp32 = sum_{b,t} dys[i][b, co, t] xs[i][b, ci, t + (k - 1) dil] for all branches of a Res2
    block in one launch (bf16 MFMA on the resident operands: exact products, fp32 sums)."""
    n = len(xs)
    if n == 0 or len(dys) != n or len(outs) != n:
        raise _hip.AirError("conv_tap_wgrad: %d inputs, %d gradients, %d outputs" % (n, len(dys), len(outs)))
    B, W, Tp = xs[0].shape
    xp, xb, yp, yb, op = [], [], [], [], []
    for x, dy, o in zip(xs, dys, outs):
        if tuple(x.shape) != (B, W, Tp) or tuple(dy.shape) != (B, W, Tp) or tuple(o.shape) != (W, W, 3):
            raise _hip.AirError("conv_tap_wgrad: branch shapes differ")
        if o.dtype != torch.float32 or not o.is_contiguous() or not o.is_cuda:
            raise _hip.AirError("conv_tap_wgrad: outputs must be contiguous fp32 GPU tensors")
        p, s = hv(x)
        xp.append(p.value), xb.append(s)
        p, s = hv(dy)
        yp.append(p.value), yb.append(s)
        op.append(o.data_ptr())
    lib = _hip.lib()
    nbytes = int(lib.air_h_conv1d_tap_wgrad_ws_bytes(ci(n), ci(B), ci(W)))
    if nbytes == 0:
        raise _hip.AirError("conv_tap_wgrad: unsupported width %d" % W)
    ws = ops.workspace(nbytes, xs[0].device)
    VP, SZ = ctypes.c_void_p * n, ctypes.c_size_t * n
    _hip.check(lib.air_h_conv1d_tap_wgrad(ci(n), ci(B), ci(W), ci(T), ci(Tp), ci(dil), VP(*xp), SZ(*xb), VP(*yp), SZ(*yb),
                                          VP(*op), dptr(ws, torch.uint8), csz(nbytes), stream()), "air_h_conv1d_tap_wgrad")
    return outs


@functools.lru_cache(maxsize=None)
def _bn_ws_bytes(B, C):
    return int(_hip.lib().air_h_bn_ws_bytes(ci(B), ci(C)))


@functools.lru_cache(maxsize=None)
def _pointwise_ws_bytes(Cout, Cin):
    lib = _hip.lib()
    return max(int(lib.air_h_conv1d_ws_bytes(ci(Cout), ci(Cin))), int(lib.air_h_conv1d_ws_bytes(ci(Cin), ci(Cout))))


@functools.lru_cache(maxsize=None)
def _wgrad_ws_bytes(B, Cin, T, Cout):
    d = _hip.AirConv1d(B, Cin, T, Cout, 1, 1, 0, 0, 0)
    return int(_hip.lib().air_conv1d_bf16_ws_bytes(ctypes.byref(d)))


def _bn_ws(B, C, device):
    n = _bn_ws_bytes(B, C)
    return ops.workspace(n, device), n


def bn_stats(x, T, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, stats_in=None):
    """(mean, invstd, scale, shift) of a resident tensor; updates the running statistics in place.  stats_in: the
    records ``conv_tap(..., stats=True)`` returned for THIS x (merged in fp64 instead of reading x)."""
    B, C, Tp = x.shape
    dev = x.device
    mean, invstd, scale, shift = torch.empty((4, C), device=dev, dtype=torch.float32).unbind(0)  # one allocation
    ws, n = _bn_ws(B, C, dev)
    p, bs = hv(x)
    _hip.check(_hip.lib().air_h_bn_stats_ex(p, csz(bs), ci(B), ci(C), ci(T), ci(Tp),
                                            dptr(stats_in, torch.uint8, allow_none=True),
                                            csz(0 if stats_in is None else stats_in.numel()), dptr(gamma), dptr(beta),
                                            cf(eps), cf(momentum), dptr(running_mean, allow_none=True),
                                            dptr(running_var, allow_none=True), dptr(mean), dptr(invstd), dptr(scale),
                                            dptr(shift), dptr(ws, torch.uint8), csz(n), stream()), "air_h_bn_stats_ex")
    return mean, invstd, scale, shift


def bn_apply(x, T, scale, shift, out=None, rowmean=None):
    B, C, Tp = x.shape
    if out is None:
        out = torch.empty((B, C, Tp), device=x.device, dtype=torch.int16)
    p, bs = hv(x)
    q, qs = hv(out)
    _hip.check(_hip.lib().air_h_bn_apply(p, csz(bs), ci(B), ci(C), ci(T), ci(Tp), dptr(scale), dptr(shift), q, csz(qs),
                                         dptr(rowmean, allow_none=True), stream()), "air_h_bn_apply")
    return out


def bn_bwd(x, dy, T, mean, invstd, gamma, dgamma, dbeta, dx=None, dy2=None, rowbias=None, rowbias_scale=1.0,
           dbias=None, relu_in=True, sums_in=None, apply=True):
    """sums_in: the records ``conv_tap(..., dgrad=True, bn=...)`` returned for THIS BatchNorm (no first pass).
    apply=False (round 6): dgamma / dbeta / dbias only - dx is computed by the consumer (``bn_bwd_prologue``)."""
    B, C, Tp = x.shape
    if dx is None and apply:
        dx = torch.empty((B, C, Tp), device=x.device, dtype=torch.int16)
    ws, n = _bn_ws(B, C, x.device)
    xp, xb = hv(x)
    gp, gb = hv(dy)
    g2p, g2b = hv(dy2, True)
    dp, db = hv(dx if apply else None, True)
    _hip.check(_hip.lib().air_h_bn_bwd_ex(xp, csz(xb), gp, csz(gb), g2p, csz(g2b), dptr(rowbias, allow_none=True),
                                          cf(rowbias_scale), ci(B), ci(C), ci(T), ci(Tp), dptr(mean), dptr(invstd),
                                          dptr(gamma), ci(1 if relu_in else 0), dp, csz(db), dptr(dgamma), dptr(dbeta),
                                          dptr(dbias, allow_none=True), dptr(sums_in, torch.uint8, allow_none=True),
                                          csz(0 if sums_in is None else sums_in.numel()), dptr(ws, torch.uint8), csz(n),
                                          stream()), "air_h_bn_bwd_ex")
    return dx


def res2_bn_apply(x, T, scale, shift, y1, add=None, y2=None):
    B, C, Tp = x.shape
    xp, xb = hv(x)
    y1p, y1b = hv(y1)
    ap, ab = hv(add, True)
    y2p, y2b = hv(y2, True)
    _hip.check(_hip.lib().air_h_res2_bn_apply(xp, csz(xb), ci(B), ci(C), ci(T), ci(Tp), dptr(scale), dptr(shift), y1p,
                                              csz(y1b), ap, csz(ab), y2p, csz(y2b), stream()), "air_h_res2_bn_apply")
    return y2


def se_scale_fwd(x, z, res, T, out):
    B, C, Tp = x.shape
    xp, xb = hv(x)
    rp, rb = hv(res)
    op, ob = hv(out)
    _hip.check(_hip.lib().air_h_se_scale_fwd(xp, csz(xb), dptr(z), rp, csz(rb), ci(B), ci(C), ci(T), ci(Tp), op, csz(ob),
                                             stream()), "air_h_se_scale_fwd")
    return out


def se_scale_bwd(x, z, dout, T):
    B, C, Tp = x.shape
    dx = torch.empty((B, C, Tp), device=x.device, dtype=torch.int16)
    dz = torch.empty((B, C), device=x.device, dtype=torch.float32)
    xp, xb = hv(x)
    dp, db = hv(dout)
    qp, qb = hv(dx)
    _hip.check(_hip.lib().air_h_se_scale_bwd(xp, csz(xb), dptr(z), dp, csz(db), ci(B), ci(C), ci(T), ci(Tp), qp, csz(qb),
                                             dptr(dz), stream()), "air_h_se_scale_bwd")
    return dx, dz


def row_stats(x, T, want_std=True, clamp_min=1e-4):
    B, C, Tp = x.shape
    mean = torch.empty((B, C), device=x.device, dtype=torch.float32)
    std = torch.empty((B, C), device=x.device, dtype=torch.float32) if want_std else None
    _hip.check(_hip.lib().air_h_row_stats(dptr(x, torch.int16), ci(B), ci(C), ci(T), ci(Tp), dptr(mean),
                                          dptr(std, allow_none=True), cf(clamp_min), stream()), "air_h_row_stats")
    return mean, std


def row_stats_bwd(x, T, mean, std, dmean, dstd, dx, accumulate=True, clamp_min=1e-4, relu_mask=False, rowsum=None):
    B, C, Tp = x.shape
    _hip.check(_hip.lib().air_h_row_stats_bwd(dptr(x, torch.int16), ci(B), ci(C), ci(T), ci(Tp), dptr(mean),
                                              dptr(std, allow_none=True), dptr(dmean, allow_none=True),
                                              dptr(dstd, allow_none=True), cf(clamp_min), dptr(dx, torch.int16),
                                              ci(1 if accumulate else 0), ci(1 if relu_mask else 0),
                                              dptr(rowsum, allow_none=True), stream()), "air_h_row_stats_bwd")
    return dx


def asp_fwd(x, logits, T):
    """logits (resident) is overwritten with the softmax weights.  Returns (B, 2C) [mu | sg] fp32."""
    B, C, Tp = x.shape
    out = torch.empty((B, 2 * C), device=x.device, dtype=torch.float32)
    _hip.check(_hip.lib().air_h_asp_fwd(dptr(x, torch.int16), dptr(logits, torch.int16), ci(B), ci(C), ci(T), ci(Tp),
                                        dptr(out), stream()), "air_h_asp_fwd")
    return out


def asp_bwd(x, w, T, out, dout, dx, rowsum=None):
    """w (resident) is overwritten with d(logits); dx (resident) is written."""
    B, C, Tp = x.shape
    _hip.check(_hip.lib().air_h_asp_bwd(dptr(x, torch.int16), dptr(w, torch.int16), ci(B), ci(C), ci(T), ci(Tp), dptr(out),
                                        dptr(dout), dptr(dx, torch.int16), dptr(rowsum, allow_none=True), stream()),
               "air_h_asp_bwd")
    return dx
