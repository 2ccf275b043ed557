"""Oracle (test infrastructure): ECAPA-TDNN (``Res2Net2``) restated as pure
functions over a parameter dict (PyTorch-CPU fp32).

Follows ecapa_tdnn.py:

* ``SEModule.forward`` ....... ecapa_tdnn.py:27-29 (layers :18-25)
* ``Bottle2neck.forward`` .... ecapa_tdnn.py:64-95 (shapes :33-62)
* ``Res2Net2.__init__`` ...... ecapa_tdnn.py:99-150
* ``Res2Net2.forward`` ....... ecapa_tdnn.py:152-198

Ordering is conv -> ReLU -> BN throughout.  Parameter names
are the reference's ``state_dict`` keys.

``bf16=True`` restates BASELINE.json configs[2] ("ECAPA-TDNN-512 bf16 train"; the reference
itself is fp32 only): the pointwise layers that hold the FLOPs - Bottle2neck conv1/conv3
(:39,:55), layer4 (:118), attention.0's layer4 part and attention.3 (:140,:143) - compute like
torch.autocast(bfloat16): both operands of the forward, dgrad and wgrad contractions are rounded
to bf16 (nearest even), products accumulate in fp32; the dilated K=3 convs of the Res2 branches (:46)
do the same in their forward and input-gradient contractions (their weight gradient stays fp32);
everything else (conv1, SE, BatchNorm, pooling, biases) stays fp32.  The bf16 oracle is pinned against the reference
through the fp32 goldens at bf16 tolerance (tests/test_oracle_golden.py).

``bf16="resident"`` (round 3, the HIP path's ``compute_dtype="bf16"``) states the bf16-RESIDENT arithmetic: every
(B, C, T) activation from the first BatchNorm's output to the pooling is a bf16 tensor - the tensors
``torch.autocast(bfloat16)`` holds in bf16 for this graph - and so is every (B, C, T) gradient.  Each op reads bf16
values, computes in fp32 and rounds what it stores ONCE (nearest even).  Rounded in the forward pass: the output of
every K = 1 / dilated K = 3 conv after bias + ReLU, every BatchNorm output, the Res2 sums ``sp + spx[i]``, the SE
result ``x * gate + residual`` (one rounding; autocast rounds the product and the sum), layer4's ReLU output, the
attention logits and the softmax weights (the pooled statistics use the STORED weights).  Rounded in the backward
pass: every data gradient where a kernel stores it - conv dgrad outputs (with the residual / concat slices folded
into the same rounding), BatchNorm backward outputs, the SE gate's d(x), the three partial sums that build d(x4).
The first layer (conv1, K = 5 on the fp32 features, ecapa_tdnn.py:111) computes like the other convs, as autocast
runs it: the features and the weight rounded to bf16 as operands, fp32 accumulation, its ReLU output and its BatchNorm
output stored bf16, its weight gradient a contraction of the stored bf16 gradient with the rounded features.
NOT rounded (wider than autocast): all statistics / per-channel and per-utterance vectors (BatchNorm statistics,
SE squeeze and MLP, context mean / std, pooled mu / sg, bn5, fc6), the conv BIASES (autocast rounds them to bf16
with the other conv arguments: ``AUTOCAST_BIAS`` below), the softmax and the pooling sums (fp32 on the
stored bf16 values), parameters and parameter gradients (fp32 accumulation of bf16 operands; the K = 3 weight
gradient an fp32 contraction of the stored bf16 values).  tests/golden/make_golden_bf16.py measures the distance of
both modes to the reference run under torch.autocast on the CPU.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
BF16_MODES = (True, "resident")
# torch.autocast casts EVERY argument of conv1d to bf16, the bias included; this build (and the HIP kernels) add the
# fp32 bias to the fp32 accumulator - wider than autocast (measured on the first layer: 3 % of the stored values move
# by one bf16 ulp).  Test-only switch: True makes the resident mode's first layer round its bias like autocast, under
# which its stored BatchNorm output reproduces the reference-under-autocast BIT FOR BIT up to fp32 summation order
# (tests/golden/make_golden_bf16.py: 5 of 98,304 values) - the sharp pin of the rounding rule.
AUTOCAST_BIAS = False


def _rnd(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Round(torch.autograd.Function):
    """Identity up to bf16 rounding of the value (fwd) and / or of the gradient flowing back through it (bwd)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _rnd(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (_rnd(g) if ctx.bwd else g), None, None


def RF(x):  # the stored tensor is bf16
    return _Round.apply(x, True, False)


def RG(x):  # the gradient stored at this point is bf16
    return _Round.apply(x, False, True)


def RB(x):
    return _Round.apply(x, True, True)


class _X4Fan(torch.autograd.Function):
    """layer4's output feeds the pooling, attention.0 and the context statistics; the HIP backward builds d(x4) in
    three stored steps: bf16(pooling) -> bf16(. + attention.0's dgrad) -> bf16(. + statistics) [ReLU mask upstream]."""

    @staticmethod
    def forward(ctx, x):
        return x.clone(), x.clone(), x.clone()

    @staticmethod
    def backward(ctx, g_pool, g_att, g_stat):
        return _rnd(_rnd(_rnd(g_pool) + g_att) + g_stat)


class _SoftmaxStored(torch.autograd.Function):
    """w = bf16(softmax_T(a)); the backward differentiates at the STORED weights (asp_bwd reads them back)."""

    @staticmethod
    def forward(ctx, a):
        w = _rnd(torch.softmax(a, dim=2))
        ctx.save_for_backward(w)
        return w

    @staticmethod
    def backward(ctx, dw):
        (w,) = ctx.saved_tensors
        return w * (dw - (w * dw).sum(2, keepdim=True))


def _bn_shapes(prefix, c, out):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def _conv_shapes(prefix, cout, cin, k, out):
    out[prefix + ".weight"] = (cout, cin, k)
    out[prefix + ".bias"] = (cout,)


def ecapa_shapes(C=512, scale=8, nOut=2, n_mels=60, bottleneck=128, attn_ch=128, context=True, encoder_type="ECA"):
    s = OrderedDict()
    _conv_shapes("conv1", C, n_mels, 5, s)
    _bn_shapes("bn1", C, s)
    width = C // scale
    for li in (1, 2, 3):
        p = "layer%d" % li
        _conv_shapes(p + ".conv1", width * scale, C, 1, s)
        _bn_shapes(p + ".bn1", width * scale, s)
        for i in range(scale - 1):
            _conv_shapes(p + ".convs.%d" % i, width, width, 3, s)
        for i in range(scale - 1):
            _bn_shapes(p + ".bns.%d" % i, width, s)
        _conv_shapes(p + ".conv3", C, width * scale, 1, s)
        _bn_shapes(p + ".bn3", C, s)
        _conv_shapes(p + ".se.se.1", bottleneck, C, 1, s)
        _bn_shapes(p + ".se.se.3", bottleneck, s)
        _conv_shapes(p + ".se.se.4", C, bottleneck, 1, s)
    _conv_shapes("layer4", 1536, 3 * C, 1, s)
    _conv_shapes("attention.0", attn_ch, 1536 * (3 if context else 1), 1, s)
    _bn_shapes("attention.2", attn_ch, s)
    _conv_shapes("attention.3", 1536 if encoder_type == "ECA" else 1, attn_ch, 1, s)  # :131-134 (ASP: one weight per frame)
    _bn_shapes("bn5", 3072, s)
    s["fc6.weight"] = (256, 3072)
    s["fc6.bias"] = (256,)
    s["fc7.weight"] = (nOut, 256)
    s["fc7.bias"] = (nOut,)
    _bn_shapes("bn7", nOut, s)
    return s


def _bn(x, p, prefix, training, updates):
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        if updates is not None:
            updates[prefix + ".running_mean"] = rm2
            updates[prefix + ".running_var"] = rv2
        return y
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


class _Bf16Pointwise(torch.autograd.Function):
    """K = 1 Conv1d with bf16-rounded operands in all three contractions, fp32 accumulation."""

    @staticmethod
    def forward(ctx, x, w):
        rnd = lambda t: t.to(torch.bfloat16).to(t.dtype)
        xb, wb = rnd(x), rnd(w)
        ctx.save_for_backward(xb, wb)
        return F.conv1d(xb, wb)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        dyb = dy.to(torch.bfloat16).to(dy.dtype)
        return F.conv_transpose1d(dyb, wb), torch.einsum("bot,bit->oi", dyb, xb).unsqueeze(2)


class _Bf16Dilated(torch.autograd.Function):
    """K = 3 dilated Conv1d (ecapa_tdnn.py:46) in bf16 compute: forward and input gradient contract
    bf16-rounded operands with fp32 accumulation; the weight gradient (0.6 % of the step's FLOPs)
    stays an fp32 contraction of the unrounded tensors, as in the HIP path."""

    @staticmethod
    def forward(ctx, x, w, dilation):
        rnd = lambda t: t.to(torch.bfloat16).to(t.dtype)
        ctx.save_for_backward(x, w)
        ctx.dilation = dilation
        return F.conv1d(rnd(x), rnd(w), None, 1, dilation, dilation)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d = ctx.dilation
        rnd = lambda t: t.to(torch.bfloat16).to(t.dtype)
        dx = F.conv_transpose1d(rnd(dy), rnd(w), None, 1, d, 0, 1, d)
        dw = torch.nn.grad.conv1d_weight(x, w.shape, dy, 1, d, d)
        return dx, dw, None


class _Bf16Conv(torch.autograd.Function):
    """Conv1d of any kernel size with bf16-rounded operands in all three contractions, fp32 accumulation (the
    resident mode's first layer: the HIP path unfolds the rounded input and runs a pointwise GEMM)."""

    @staticmethod
    def forward(ctx, x, w, dilation, padding):
        xb, wb = _rnd(x), _rnd(w)
        ctx.save_for_backward(xb, wb)
        ctx.geom = (dilation, padding)
        return F.conv1d(xb, wb, None, 1, padding, dilation)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        d, pad = ctx.geom
        dyb = _rnd(dy)
        dx = torch.nn.grad.conv1d_input(xb.shape, wb, dyb, 1, pad, d) if ctx.needs_input_grad[0] else None
        return dx, torch.nn.grad.conv1d_weight(xb, wb.shape, dyb, 1, pad, d), None, None


def _conv(x, p, prefix, dilation=1, padding=0, bf16=False):
    w = p[prefix + ".weight"]
    if bf16 and w.shape[2] == 1:
        return _Bf16Pointwise.apply(x, w) + p[prefix + ".bias"][None, :, None]
    if bf16 and w.shape[2] == 3 and padding == dilation and w.shape[0] % 64 == 0 and w.shape[1] % 32 == 0:
        return _Bf16Dilated.apply(x, w, dilation) + p[prefix + ".bias"][None, :, None]
    return F.conv1d(x, w, p[prefix + ".bias"], 1, padding, dilation)


def se_module(x, p, prefix, training, updates):
    """ecapa_tdnn.py:18-29: mean over T -> 1x1 -> ReLU -> BN -> 1x1 -> sigmoid; scale."""
    s = x.mean(dim=2, keepdim=True)
    s = F.relu(_conv(s, p, prefix + ".se.1"))
    s = _bn(s, p, prefix + ".se.3", training, updates)
    s = torch.sigmoid(_conv(s, p, prefix + ".se.4"))
    return x * s


def bottle2neck_resident(x, p, prefix, dilation, scale, training, updates):
    """ecapa_tdnn.py:64-95 on bf16-resident tensors (module docstring): x is a bf16-valued tensor."""
    out = RF(_bn(RB(F.relu(_conv(x, p, prefix + ".conv1", bf16=True))), p, prefix + ".bn1", training, updates))
    width = out.shape[1] // scale
    spx = torch.split(out, width, 1)
    outs = []
    sp = None
    for i in range(scale - 1):
        sp = RG(spx[i]) if i == 0 else RB(sp + spx[i])
        sp = RB(F.relu(_conv(sp, p, prefix + ".convs.%d" % i, dilation, dilation, bf16=True)))
        sp = RF(_bn(sp, p, prefix + ".bns.%d" % i, training, updates))
        outs.append(sp)
    outs.append(spx[scale - 1])
    cat = RG(torch.cat(outs, 1))
    o3 = RF(_bn(RB(F.relu(_conv(cat, p, prefix + ".conv3", bf16=True))), p, prefix + ".bn3", training, updates))
    # SEModule (:18-29) in fp32 on the stored tensor; gate * x + residual stored with one rounding
    s = o3.mean(dim=2, keepdim=True)
    s = F.relu(_conv(s, p, prefix + ".se.se.1"))
    s = _bn(s, p, prefix + ".se.se.3", training, updates)
    s = torch.sigmoid(_conv(s, p, prefix + ".se.se.4"))
    return RB(RG(o3) * s + x)


def bottle2neck(x, p, prefix, dilation, scale, training, updates, bf16=False):
    """ecapa_tdnn.py:64-95."""
    out = _bn(F.relu(_conv(x, p, prefix + ".conv1", bf16=bf16)), p, prefix + ".bn1", training, updates)
    width = out.shape[1] // scale
    spx = torch.split(out, width, 1)
    outs = []
    sp = None
    for i in range(scale - 1):
        sp = spx[i] if i == 0 else sp + spx[i]
        sp = _conv(sp, p, prefix + ".convs.%d" % i, dilation, dilation, bf16=bf16)  # k=3: pad = dilation (:46)
        sp = _bn(F.relu(sp), p, prefix + ".bns.%d" % i, training, updates)
        outs.append(sp)
    outs.append(spx[scale - 1])
    out = torch.cat(outs, 1)
    out = _bn(F.relu(_conv(out, p, prefix + ".conv3", bf16=bf16)), p, prefix + ".bn3", training, updates)
    out = se_module(out, p, prefix + ".se", training, updates)
    return out + x


def ecapa_forward(p, x, scale=8, training=True, updates=None, taps=None, context=True, out_bn=True,
                  bf16=False, summed=False):
    """Res2Net2.forward (ecapa_tdnn.py:152-198), encoder_type 'ECA'.  ``context`` (:126-129, :177-180) and ``summed``
    (:163-170: layer2 / layer3 read x + x1 / x + x1 + x2) are the constructor options the reference's own score files
    were made with (lfcc_ecapa512c{t,f}s{t,f}_*; main_train.py:167 passes the defaults, context=True, summed=False).
    x: (B, n_mels, T).  Returns (feat (B,256), out (B,nOut))."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    if bf16 == "resident":
        assert not summed, "resident arithmetic is stated for summed=False (main_train.py:167)"
        return _ecapa_forward_resident(p, x, scale, training, updates, tap, context, out_bn)
    x = _bn(F.relu(_conv(x, p, "conv1", 1, 2)), p, "bn1", training, updates)  # :159-161
    x1 = tap("x1", bottle2neck(x, p, "layer1", 2, scale, training, updates, bf16))
    if summed:  # :163-166
        x2 = tap("x2", bottle2neck(x + x1, p, "layer2", 3, scale, training, updates, bf16))
        x3 = tap("x3", bottle2neck(x + x1 + x2, p, "layer3", 4, scale, training, updates, bf16))
    else:  # :167-170
        x2 = tap("x2", bottle2neck(x1, p, "layer2", 3, scale, training, updates, bf16))
        x3 = tap("x3", bottle2neck(x2, p, "layer3", 4, scale, training, updates, bf16))
    x = F.relu(_conv(torch.cat((x1, x2, x3), 1), p, "layer4", bf16=bf16))  # :172-173
    tap("layer4", x)
    t = x.shape[-1]
    if context:  # :177-178
        gx = torch.cat((x, x.mean(2, keepdim=True).repeat(1, 1, t),
                        torch.sqrt(x.var(2, keepdim=True).clamp(min=1e-4)).repeat(1, 1, t)), 1)
    else:
        gx = x
    if bf16 and context:
        # the time-constant mean/std rows of gx enter attention.0 as a per-utterance fp32 term;
        # only the layer4 part is a (B, 1536, T) contraction and runs in bf16
        w0, c = p["attention.0.weight"], x.shape[1]
        a0 = (_Bf16Pointwise.apply(x, w0[:, :c]) + F.conv1d(gx[:, c:, :1], w0[:, c:])
              + p["attention.0.bias"][None, :, None])
    else:
        a0 = _conv(gx, p, "attention.0", bf16=bf16)
    a = _bn(F.relu(a0), p, "attention.2", training, updates)
    w = torch.softmax(_conv(a, p, "attention.3", bf16=bf16), dim=2)  # :139-145
    tap("w", w)
    mu = torch.sum(x * w, dim=2)  # :184
    sg = torch.sqrt((torch.sum((x ** 2) * w, dim=2) - mu ** 2).clamp(min=1e-4))  # :185
    tap("mu", mu)
    tap("sg", sg)
    x = _bn(torch.cat((mu, sg), 1), p, "bn5", training, updates)  # :187-189
    feat = F.linear(x, p["fc6.weight"], p["fc6.bias"])  # :191
    out = F.linear(feat, p["fc7.weight"], p["fc7.bias"])  # :193
    if out_bn:
        out = _bn(out, p, "bn7", training, updates)  # :195-196
    return feat, out


def _ecapa_forward_resident(p, x, scale, training, updates, tap, context, out_bn):
    """Res2Net2.forward (ecapa_tdnn.py:152-198) with bf16-resident activations (module docstring)."""
    assert context, "resident arithmetic is stated for context=True (main_train.py:167)"
    b1 = _rnd(p["conv1.bias"]) if AUTOCAST_BIAS else p["conv1.bias"]
    c1 = _Bf16Conv.apply(x, p["conv1.weight"], 1, 2) + b1[None, :, None]  # :159
    h = RF(_bn(RB(F.relu(c1)), p, "bn1", training, updates))  # :160-161, stored like every other conv -> ReLU -> BN
    tap("h0", h)
    x1 = tap("x1", bottle2neck_resident(h, p, "layer1", 2, scale, training, updates))
    x2 = tap("x2", bottle2neck_resident(x1, p, "layer2", 3, scale, training, updates))
    x3 = tap("x3", bottle2neck_resident(x2, p, "layer3", 4, scale, training, updates))
    x4 = RF(F.relu(_conv(RG(torch.cat((x1, x2, x3), 1)), p, "layer4", bf16=True)))  # :172-173
    tap("layer4", x4)
    x_pool, x_att, x_stat = _X4Fan.apply(x4)
    mean = x_stat.mean(2, keepdim=True)
    std = torch.sqrt(x_stat.var(2, keepdim=True).clamp(min=1e-4))
    w0, c = p["attention.0.weight"], x4.shape[1]
    a0 = (_Bf16Pointwise.apply(x_att, w0[:, :c]) + F.conv1d(torch.cat((mean, std), 1), w0[:, c:])
          + p["attention.0.bias"][None, :, None])
    a = RF(_bn(RB(F.relu(a0)), p, "attention.2", training, updates))
    logits = RB(_conv(RG(a), p, "attention.3", bf16=True))
    w = _SoftmaxStored.apply(logits)  # :139-145
    tap("w", w)
    mu = torch.sum(x_pool * w, dim=2)  # :184
    sg = torch.sqrt((torch.sum((x_pool ** 2) * w, dim=2) - mu ** 2).clamp(min=1e-4))  # :185
    tap("mu", mu)
    tap("sg", sg)
    y = _bn(torch.cat((mu, sg), 1), p, "bn5", training, updates)  # :187-189
    feat = F.linear(y, p["fc6.weight"], p["fc6.bias"])  # :191
    out = F.linear(feat, p["fc7.weight"], p["fc7.bias"])  # :193
    if out_bn:
        out = _bn(out, p, "bn7", training, updates)  # :195-196
    return feat, out
