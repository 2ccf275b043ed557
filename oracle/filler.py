"""Oracle (test infrastructure): closed-form parameter filler.

The reference initialises weights from the torch RNG in an order that depends
on discarded sub-modules (resnet.py:162-166) and leaves the torch CPU RNG
unseeded (utils.py:27).  Instead of chasing RNG parity, goldens and tests fill
every parameter with a deterministic closed form so no large weight file has to
be committed and no init-order dependence exists.
"""
import math

import numpy as np
import torch


def _seed_of(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF  # FNV-1a
    return h


def fill_value(name, shape):
    """Deterministic float32 tensor for parameter ``name`` of ``shape``.

    Values come from numpy's PCG64 stream seeded by a hash of the name (bit-stable
    across platforms), scaled like the reference's initialiser (resnet.py:149-157:
    kaiming-normal fan_out for convs, kaiming-uniform for linears) so the network
    is as well conditioned as a freshly initialised one.  (A smooth closed form such
    as sin(0.37 i) makes the convs cancel catastrophically: fp32 and fp64 CPU
    evaluations of the same net then differ by 1e-3, which would swamp parity.)
    """
    n = int(np.prod(shape)) if len(shape) else 1
    rng = np.random.Generator(np.random.PCG64(_seed_of(name)))
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        v = 0.05 * rng.standard_normal(n)
    elif leaf == "running_var":
        v = 1.0 + 0.2 * rng.uniform(-1, 1, n)
    elif len(shape) >= 3:  # conv weight (Cout, Cin, k...): kaiming normal, fan_out
        fan_out = shape[0] * int(np.prod(shape[2:]))
        v = math.sqrt(2.0 / fan_out) * rng.standard_normal(n)
    elif len(shape) == 2:  # linear weight / attention vector / loss centre: uniform, fan_in
        bound = math.sqrt(3.0 / shape[1])
        v = rng.uniform(-bound, bound, n)
    elif leaf == "weight":  # BN gamma
        v = 1.0 + 0.1 * rng.uniform(-1, 1, n)
    else:  # biases, BN beta
        v = 0.1 * rng.uniform(-1, 1, n)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fill_state(shapes):
    """shapes: dict name -> shape.  Returns dict name -> tensor."""
    return {k: fill_value(k, tuple(s)) for k, s in shapes.items()}


def fill_module_(module):
    """Fill a torch module's parameters and buffers in place by name."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(fill_value(k, tuple(v.shape)).to(v.dtype))
    return module


def synth_pcm(B, L, seed):
    """0.1 * N(0,1) float32 PCM (SURVEY.md §8d synthetic input)."""
    g = torch.Generator().manual_seed(seed)
    return 0.1 * torch.randn(B, L, generator=g, dtype=torch.float32)


def synth_feat(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=g, dtype=torch.float32)
