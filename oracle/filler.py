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


def fill_value(name, shape):
    """Deterministic float32 tensor for parameter ``name`` of ``shape``.

    p.flat[i] = gain * sin(0.37 i + phase(name)); gain follows the fan-in/out
    scale the reference's initialiser would give (resnet.py:149-157) so
    activations stay O(1) through the stack.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    phase = (sum(ord(c) * (k + 1) for k, c in enumerate(name)) % 997) * 0.013
    i = np.arange(n, dtype=np.float64)
    base = np.sin(0.37 * i + phase)
    leaf = name.split(".")[-1]
    if leaf == "running_mean":
        v = 0.05 * base
    elif leaf == "running_var":
        v = 1.0 + 0.2 * base
    elif leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    elif len(shape) >= 3:  # conv weight (Cout, Cin, k...) : kaiming fan_out scale
        fan_out = shape[0] * int(np.prod(shape[2:]))
        v = math.sqrt(2.0) * math.sqrt(2.0 / fan_out) * base
        # ECAPA convs are followed by ReLU->BN (no BN before): use fan_in so they stay O(1)
    elif len(shape) == 2:  # linear weight / attention vector / loss centre
        fan_in = shape[1]
        v = math.sqrt(2.0) * math.sqrt(1.0 / fan_in) * base
    elif leaf == "weight":  # BN gamma
        v = 1.0 + 0.1 * base
    else:  # biases, BN beta
        v = 0.1 * base
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
