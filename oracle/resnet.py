"""Oracle (test infrastructure): ResNet-18 (pre-activation) + SelfAttention
pooling restated as pure functions over a parameter dict (PyTorch-CPU fp32).

Follows resnet.py (identical to the model.py copies):

* ``SelfAttention.forward`` .... resnet.py:23-46
* ``PreActBlock.forward`` ...... resnet.py:63-69
* ``ResNet.__init__`` shapes ... resnet.py:123-147, ``_make_layer`` :159-172
* ``ResNet.forward`` ........... resnet.py:174-191

Parameter names are the reference's ``state_dict`` keys so a reference
checkpoint maps one-to-one.  Backward comes from torch autograd on the CPU.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYER_PLANES = (64, 128, 256, 512)
LAYER_STRIDES = (1, 2, 2, 2)
BLOCKS_PER_LAYER = 2  # RESNET_CONFIGS['18'] (resnet.py:103)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn_shapes(prefix, c, out):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def resnet18_shapes(num_nodes=3, enc_dim=256, nclasses=2):
    """state_dict key -> shape, in the reference's registration order
    (resnet.py:131-147; ``attention`` is created last, :147)."""
    s = OrderedDict()
    s["conv1.weight"] = (16, 1, 9, 3)
    _bn_shapes("bn1", 16, s)
    in_planes = 16
    for li, (planes, stride) in enumerate(zip(LAYER_PLANES, LAYER_STRIDES), start=1):
        for bi in range(BLOCKS_PER_LAYER):
            p = "layer%d.%d" % (li, bi)
            st = stride if bi == 0 else 1
            _bn_shapes(p + ".bn1", in_planes, s)
            s[p + ".conv1.weight"] = (planes, in_planes, 3, 3)
            _bn_shapes(p + ".bn2", planes, s)
            s[p + ".conv2.weight"] = (planes, planes, 3, 3)
            if st != 1 or in_planes != planes:
                s[p + ".shortcut.0.weight"] = (planes, in_planes, 1, 1)
            in_planes = planes
    s["conv5.weight"] = (256, 512, num_nodes, 3)
    _bn_shapes("bn5", 256, s)
    s["fc.weight"] = (enc_dim, 512)
    s["fc.bias"] = (enc_dim,)
    s["fc_mu.weight"] = (nclasses if nclasses >= 2 else 1, enc_dim)
    s["fc_mu.bias"] = (nclasses if nclasses >= 2 else 1,)
    s["attention.att_weights"] = (1, 256)
    return s


def is_buffer(name):
    leaf = name.split(".")[-1]
    return leaf in ("running_mean", "running_var", "num_batches_tracked")


def _bn(x, p, prefix, training, updates):
    """nn.BatchNorm2d forward; running-stat updates are returned, not applied."""
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


class ReluProbe:
    """Test helper (parity of the ReLU decisions, tests/test_resnet_gpu.py): stands in for F.relu in the forward
    below.  ``masks`` = None: records the sign decision of every ReLU in execution order.  ``masks`` = a list of
    boolean tensors (the decisions ANOTHER evaluation of the same net took, e.g. the HIP path's): applies those
    instead of its own - y = x * mask, so the backward pass gates with them too - and counts per ReLU where its
    own sign would have differed (``flips``) together with the largest |pre-activation| among them (``flip_mag``);
    ``scale`` / ``count``: the largest |pre-activation| and the number of units of every ReLU."""

    def __init__(self, masks=None):
        self.masks, self.own, self.flips, self.flip_mag = masks, [], [], []
        self.scale, self.count = [], []  # per ReLU: max |pre-activation|, number of units

    def __call__(self, x):
        own = x.detach() > 0
        i = len(self.own)
        self.own.append(own)
        self.scale.append(float(x.detach().abs().max()))
        self.count.append(int(x.numel()))
        if self.masks is None:
            return F.relu(x)
        m = self.masks[i].to(device=x.device).reshape(x.shape)
        diff = own != m
        self.flips.append(int(diff.sum()))
        self.flip_mag.append(float(x.detach().abs()[diff].max()) if bool(diff.any()) else 0.0)
        return x * m.to(x.dtype)


def preact_block(x, p, prefix, stride, training, updates, relu=F.relu):
    """resnet.py:63-69.  The 1x1 shortcut acts on the ACTIVATED tensor (:64-65)."""
    out = relu(_bn(x, p, prefix + ".bn1", training, updates))
    key = prefix + ".shortcut.0.weight"
    shortcut = F.conv2d(out, p[key], None, stride) if key in p else x
    out = F.conv2d(out, p[prefix + ".conv1.weight"], None, stride, 1)
    out = F.conv2d(relu(_bn(out, p, prefix + ".bn2", training, updates)),
                   p[prefix + ".conv2.weight"], None, 1, 1)
    return out + shortcut


def self_attention_pool(x, att_weights, noise=None, mean_only=False):
    """resnet.py:23-46.  x: (B, T, H).  ``noise`` is the (B, T, H) tensor the
    reference draws as ``1e-5*torch.randn`` on the HOST every call (:38);
    pass it explicitly (already scaled) or None for no noise."""
    w = x @ att_weights.reshape(-1, 1)  # (B, T, 1)  (:26)
    att = torch.softmax(torch.tanh(w.squeeze(2)), dim=1)  # (:28-33)
    weighted = x * att.unsqueeze(2)
    avg = weighted.sum(1)
    if mean_only:
        return avg
    z = weighted if noise is None else weighted + noise
    std = z.std(1)  # unbiased (:42)
    return torch.cat((avg, std), 1)


def resnet18_forward(p, x, training=True, noise=None, updates=None, taps=None, relu=None):
    """ResNet.forward (resnet.py:174-191).

    p: dict of tensors keyed like the reference state_dict.
    x: (B, 1, 60, T).  Returns (feat (B, enc_dim), mu (B, nclasses)).
    ``updates`` (dict) receives new BN running stats when training.
    ``taps`` (dict) receives intermediate activations for layer-wise parity.
    ``relu``: stand-in for F.relu (a ReluProbe); the 18 ReLUs run in the order stem, (bn1, bn2) per block, bn5.
    """
    relu = F.relu if relu is None else relu
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    x = F.conv2d(x, p["conv1.weight"], None, (3, 1), (1, 1))  # :176
    tap("conv1", x)
    x = relu(_bn(x, p, "bn1", training, updates))  # :177
    for li, stride in enumerate(LAYER_STRIDES, start=1):
        for bi in range(BLOCKS_PER_LAYER):
            x = preact_block(x, p, "layer%d.%d" % (li, bi), stride if bi == 0 else 1,
                             training, updates, relu)
        tap("layer%d" % li, x)
    x = F.conv2d(x, p["conv5.weight"], None, 1, (0, 1))  # :182
    tap("conv5", x)
    x = relu(_bn(x, p, "bn5", training, updates)).squeeze(2)  # :183
    stats = self_attention_pool(x.permute(0, 2, 1).contiguous(),
                                p["attention.att_weights"], noise)  # :185
    tap("stats", stats)
    feat = F.linear(stats, p["fc.weight"], p["fc.bias"])  # :187
    mu = F.linear(feat, p["fc_mu.weight"], p["fc_mu.bias"])  # :189
    return feat, mu
