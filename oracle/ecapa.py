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
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn_shapes(prefix, c, out):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def _conv_shapes(prefix, cout, cin, k, out):
    out[prefix + ".weight"] = (cout, cin, k)
    out[prefix + ".bias"] = (cout,)


def ecapa_shapes(C=512, scale=8, nOut=2, n_mels=60, bottleneck=128, attn_ch=128, context=True):
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
    _conv_shapes("attention.3", 1536, attn_ch, 1, s)
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
                  bf16=False):
    """Res2Net2.forward (ecapa_tdnn.py:152-198), encoder_type 'ECA', summed=False.
    x: (B, n_mels, T).  Returns (feat (B,256), out (B,nOut))."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = _bn(F.relu(_conv(x, p, "conv1", 1, 2)), p, "bn1", training, updates)  # :159-161
    x1 = tap("x1", bottle2neck(x, p, "layer1", 2, scale, training, updates, bf16))
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
