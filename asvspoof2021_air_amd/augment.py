"""On-the-fly channel (impulse-response) augmentation in the HIP front-end
(BASELINE.json configs[4]).

The reference augments its corpora offline with idiap/acoustic-simulator
(channel_simulation/simulated_device.py:16-61: one random device IR per utterance, written back
as wav files).  Here the same operation runs on the GPU between the PCM batch and the LFCC
kernel: ``y = (x * h)[:L]`` rescaled to the input peak, one IR per utterance drawn from a bank
(``random.choice`` semantics, seeded), with probability ``p`` per utterance.  The tool's ``.ir``
files are not distributable with the reference, so ``synthetic_ir_bank`` provides device-like
(short, coloured) and room-like (exponentially decaying noise tail) responses; a real bank loads
with ``ChannelAugment(irs=tensor)``.  Arithmetic spec and parity status: oracle/channel.py.
"""
import ctypes

import numpy as np
import torch

from . import _hip, ops


def synthetic_ir_bank(n_device=27, n_space=3, taps=1024, sr=16000, seed=688):
    """(n_device + n_space, taps) float32.  Counts follow simulated_device.py:38-39
    (``random.sample(recDevices, 27)``, ``random.sample(recSpace, 3)``).
    Device IRs: direct path + a few early taps + a two-pole resonance within ~4 ms;
    space IRs: direct path + exponentially decaying noise, RT60 in 0.15-0.5 s (cut at ``taps``)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bank = np.zeros((n_device + n_space, taps), dtype=np.float64)
    n = np.arange(taps)
    for i in range(n_device):
        h = np.zeros(taps)
        h[0] = 1.0
        k = min(taps, 64)
        fc, bw = rng.uniform(300.0, 6000.0), rng.uniform(200.0, 2000.0)
        r = np.exp(-np.pi * bw / sr)
        res = (r ** n[:k]) * np.cos(2 * np.pi * fc / sr * n[:k] + rng.uniform(0, np.pi))
        h[:k] += rng.uniform(0.2, 0.9) * res
        h[1:k] += 0.05 * rng.standard_normal(k - 1) * np.exp(-n[1:k] / 8.0)
        bank[i] = h
    for i in range(n_space):
        rt60 = rng.uniform(0.15, 0.5)
        decay = np.exp(-6.907755 * n / (rt60 * sr))
        h = 0.3 * rng.standard_normal(taps) * decay
        h[: int(0.002 * sr)] *= 0.1
        h[0] = 1.0
        bank[n_device + i] = h
    bank /= np.sqrt((bank ** 2).sum(1, keepdims=True))
    return torch.from_numpy(bank.astype(np.float32))


def ir_convolve(pcm, irs, idx=None, normalize=True, out=None):
    """pcm (B, L) fp32 GPU, irs (n_ir, H) fp32 GPU, idx (B,) int32 GPU or None -> (B, L)."""
    if not pcm.is_cuda or not irs.is_cuda:
        raise _hip.AirError("ir_convolve needs GPU tensors; there is no CPU fallback")
    B, L = pcm.shape
    n_ir, H = irs.shape
    y = out if out is not None else torch.empty_like(pcm)
    lib = _hip.lib()
    n = lib.air_ir_convolve_ws_bytes_ex(_hip.ci(B), _hip.ci(n_ir), _hip.ci(H))  # (+ the FFT tables when H qualifies)
    ws = ops.workspace(n, pcm.device)
    _hip.check(lib.air_ir_convolve(_hip.dptr(pcm), _hip.ci(B), _hip.ci(L), _hip.dptr(irs), _hip.ci(n_ir),
                                   _hip.ci(H), _hip.dptr(idx, torch.int32, True), _hip.ci(1 if normalize else 0),
                                   _hip.dptr(y), _hip.dptr(ws, torch.uint8), _hip.csz(n), _hip.stream()),
               "air_ir_convolve")
    return y


class ChannelAugment:
    """Per-utterance random IR from a bank, applied with probability ``p``."""

    def __init__(self, irs=None, p=1.0, seed=688, normalize=True, device="cuda"):
        self.irs = (irs if irs is not None else synthetic_ir_bank()).to(device=device, dtype=torch.float32).contiguous()
        self.p = float(p)
        self.normalize = normalize
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self._pinned = {}  # batch size -> pinned host staging buffer for the index upload

    def draw(self, batch):
        """(B,) int32 IR indices, -1 = leave the utterance unchanged."""
        idx = self.rng.integers(0, self.irs.shape[0], size=batch).astype(np.int32)
        if self.p < 1.0:
            idx[self.rng.random(batch) >= self.p] = -1
        return idx

    def __call__(self, pcm, idx=None):
        if idx is None:
            idx = self.draw(pcm.shape[0])
        if not torch.is_tensor(idx):
            # asynchronous upload from a pinned buffer: a pageable .to(device) would synchronise the
            # stream and drain the step's launch queue (measured: ~1 ms per step)
            idx = np.asarray(idx, dtype=np.int32)
            slot = self._pinned.get(idx.size)
            if slot is None:
                slot = self._pinned[idx.size] = [torch.empty(idx.size, dtype=torch.int32).pin_memory(), None]
            if slot[1] is not None:
                slot[1].synchronize()  # the previous upload from this buffer has executed (one step of run-ahead)
            slot[0].copy_(torch.from_numpy(idx))
            idx = slot[0].to(pcm.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return ir_convolve(pcm, self.irs, idx.to(device=pcm.device, dtype=torch.int32).contiguous(), self.normalize)
