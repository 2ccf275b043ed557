"""Drop-in for the reference's ``feature_extraction.LFCC`` (feature_extraction.py:61-138).

Same constructor, same ``forward(x:(B,L)) -> (B, 1+L//fs, 3*filter_num)``,
same ``state_dict`` keys (``lfcc_fb``, ``l_dct.weight``), same in-place
pre-emphasis of the caller's tensor (:106).  The device work is ONE fused HIP
kernel (csrc/lfcc.hip) reached through ``air_lfcc_fwd``.
"""
import math

import torch
import torch.nn as nn

from . import _hip


def trimf(x, params):
    """Triangular membership (feature_extraction.py:16-39): strict inequalities on
    both slopes, exactly 1 at x == b."""
    if len(params) != 3:
        raise ValueError("trimf requires params to be a list of 3 elements")
    a, b, c = params
    if a > b or b > c:
        raise ValueError("trimf(x, [a, b, c]) requires a<=b<=c")
    y = torch.zeros_like(x, dtype=torch.float32)
    if a < b:
        up = (a < x) & (x < b)
        y[up] = (x[up] - a) / (b - a)
    if b < c:
        down = (b < x) & (x < c)
        y[down] = (c - x[down]) / (c - b)
    y[x == b] = 1
    return y


def dct_ortho(x):
    """Orthonormal DCT-II over the last dim via an FFT of the even/odd-reordered
    signal (the algorithm of utils_dsp.py:147-176, on torch.fft)."""
    n = x.shape[-1]
    flat = x.contiguous().view(-1, n)
    v = torch.cat([flat[:, ::2], flat[:, 1::2].flip([1])], dim=1)
    vc = torch.view_as_real(torch.fft.fft(v))
    k = -torch.arange(n, dtype=x.dtype)[None, :] * math.pi / (2 * n)
    out = vc[:, :, 0] * torch.cos(k) - vc[:, :, 1] * torch.sin(k)
    out[:, 0] /= math.sqrt(n) * 2
    out[:, 1:] /= math.sqrt(n / 2) * 2
    return 2 * out.view(*x.shape)


class LinearDCT(nn.Linear):
    """DCT as a frozen linear layer (utils_dsp.py:220-244); only type 'dct' is on the path."""

    def __init__(self, in_features, type, norm=None, bias=False):
        if type != "dct" or norm != "ortho":
            raise NotImplementedError("only LinearDCT(n, 'dct', norm='ortho') is on the hot path")
        self.type, self.N, self.norm = type, in_features, norm
        super().__init__(in_features, in_features, bias=bias)

    def reset_parameters(self):
        self.weight.data = dct_ortho(torch.eye(self.N)).data.t().contiguous()
        self.weight.requires_grad = False


def delta(x):
    """feature_extraction.py:41-58 on the host (kept for API parity; the kernel fuses it)."""
    xp = torch.cat((x[:, :1], x, x[:, -1:]), 1)
    return xp[:, 2:] - xp[:, :-2]


PAD_MODES = {"repeat": 0, "zero": 1, "silence": 2}


def pad_mode_id(padding):
    """--padding choice -> AIR_PAD_* (include/air_hip.h); the reference's error for anything else (dataset.py:79)."""
    if padding not in PAD_MODES:
        raise ValueError("Padding should be zero or repeat!")
    return PAD_MODES[padding]


class LFCC(nn.Module):
    """LFCC(fl, fs, fn, sr, filter_num, with_energy=False, with_emphasis=True, with_delta=True)."""

    def __init__(self, fl, fs, fn, sr, filter_num, with_energy=False, with_emphasis=True,
                 with_delta=True):
        super().__init__()
        self.fl, self.fs, self.fn, self.sr, self.filter_num = fl, fs, fn, sr, filter_num
        f = (sr / 2) * torch.linspace(0, 1, fn // 2 + 1)
        bands = torch.linspace(min(f), max(f), filter_num + 2)
        fb = torch.zeros([fn // 2 + 1, filter_num])
        for idx in range(filter_num):
            fb[:, idx] = trimf(f, [bands[idx], bands[idx + 1], bands[idx + 2]])
        self.lfcc_fb = nn.Parameter(fb, requires_grad=False)
        self.l_dct = LinearDCT(filter_num, "dct", norm="ortho")
        self.with_energy = with_energy
        self.with_emphasis = with_emphasis
        self.with_delta = with_delta
        self.mutate_input = True  # reference behaviour (feature_extraction.py:106)
        self._plan = None
        self._plan_key = None

    @property
    def out_dim(self):
        return self.filter_num * (3 if self.with_delta else 1)

    def _flags(self):
        return (1 if self.with_emphasis else 0) | (2 if self.with_delta else 0)

    def plan(self, device):
        """Device-resident plan blob (window, twiddles, sparse filterbank, DCT)."""
        key = (str(device), self.lfcc_fb._version, self.l_dct.weight._version,
               self.lfcc_fb.data_ptr(), self.l_dct.weight.data_ptr())
        if self._plan is None or self._plan_key != key:
            L = _hip.lib()
            nbytes = L.air_lfcc_plan_bytes()
            host = torch.zeros(nbytes, dtype=torch.uint8)
            fb = self.lfcc_fb.detach().float().cpu().contiguous()
            dct = self.l_dct.weight.detach().float().cpu().contiguous()
            win = torch.hamming_window(self.fl).contiguous()
            _hip.check(L.air_lfcc_plan_build(_hip.hptr(fb), _hip.ci(fb.shape[0]), _hip.ci(fb.shape[1]),
                                             _hip.hptr(dct), _hip.hptr(win), _hip.ci(self.fl),
                                             _hip.ci(self.fs), _hip.ci(self.fn),
                                             _hip.hptr(host, torch.uint8)), "air_lfcc_plan_build")
            self._plan = host.to(device)
            self._plan_key = key
        return self._plan

    def _check(self, x):
        if self.with_energy:
            raise NotImplementedError("with_energy=True is not on the hot path (dead branch, "
                                      "feature_extraction.py:123-127)")
        if x.dim() != 2:
            raise ValueError("LFCC expects (batch, length), got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise _hip.AirError("LFCC HIP path needs a GPU tensor; there is no CPU fallback")

    def forward(self, x):
        self._check(x)
        B, L = x.shape
        if x.dtype == torch.int16:  # raw 16-bit PCM: nothing to mutate (the reference never sees integers)
            return self._forward_i16(x, 0, None)
        T = 1 + L // self.fs
        src = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        out = torch.empty((B, T, self.out_dim), device=x.device, dtype=torch.float32)
        lib = _hip.lib()
        _hip.check(lib.air_lfcc_fwd(_hip.dptr(src), _hip.ci(B), _hip.ci(L), _hip.dptr(out),
                                    _hip.dptr(self.plan(x.device), torch.uint8),
                                    _hip.ci(self._flags()), _hip.stream()), "air_lfcc_fwd")
        if self.with_emphasis and self.mutate_input:
            self._preemph_inplace(src)
            if src is not x:
                x.copy_(src)
        return out

    def silence_row(self, device):
        """The frame the reference prepends under ``--padding silence``: frame 0 of the LFCC of 3200 zero
        samples (dataset.py:13-16), computed once per device by this module's own kernel."""
        key = (str(device), self._plan_key)
        if getattr(self, "_silence", None) is None or self._silence_key != key:
            mut, self.mutate_input = self.mutate_input, False
            try:
                row = self.forward(torch.zeros(1, 3200, device=device))[0, 0].contiguous()
            finally:
                self.mutate_input = mut
            self._silence, self._silence_key = row, (str(device), self._plan_key)
        return self._silence

    def forward_padded(self, x, feat_len=750, start=None, padding="repeat"):
        """Fused LFCC -> pad/chop -> transpose: (B, L) -> (B, out_dim, feat_len), i.e. what
        dataset.py:66-79 + main_train.py:338 build on the host.  Does not mutate ``x``.
        ``start``: optional int32 (B,) crop offsets for T > feat_len.  ``padding``: the reference's
        ``--padding`` choices 'repeat' | 'zero' | 'silence' (main_train.py:45); anything else raises
        ValueError like dataset.py:79."""
        self._check(x)
        mode = pad_mode_id(padding)
        B, L = x.shape
        i16 = x.dtype == torch.int16
        src = x.contiguous() if i16 else (x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous())
        out = torch.empty((B, self.out_dim, feat_len), device=x.device, dtype=torch.float32)
        sil = self.silence_row(x.device) if mode == 2 else None
        lib = _hip.lib()
        null = _hip.dptr(None, allow_none=True)
        _hip.check(lib.air_lfcc_fwd_padded_ex(null if i16 else _hip.dptr(src),
                                              _hip.dptr(src, torch.int16) if i16 else null,
                                              _hip.ci(B), _hip.ci(L), _hip.dptr(out),
                                              _hip.ci(feat_len), _hip.dptr(start, torch.int32, True),
                                              _hip.dptr(self.plan(x.device), torch.uint8),
                                              _hip.ci(self._flags()), _hip.ci(mode), _hip.dptr(sil, allow_none=True),
                                              _hip.stream()),
                   "air_lfcc_fwd_padded_ex")
        return out

    def _forward_i16(self, x, feat_len, start):
        """16-bit PCM straight from the wav/flac samples (x = s / 32768 inside the kernel: the floats
        soundfile / librosa hand the reference), in either layout (feat_len 0 = (B, T, D))."""
        B, L = x.shape
        x = x.contiguous()
        T = 1 + L // self.fs
        shape = (B, self.out_dim, feat_len) if feat_len > 0 else (B, T, self.out_dim)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
        lib = _hip.lib()
        _hip.check(lib.air_lfcc_fwd_padded_i16(_hip.dptr(x, torch.int16), _hip.ci(B), _hip.ci(L), _hip.dptr(out),
                                               _hip.ci(feat_len), _hip.dptr(start, torch.int32, True),
                                               _hip.dptr(self.plan(x.device), torch.uint8),
                                               _hip.ci(self._flags()), _hip.stream()),
                   "air_lfcc_fwd_padded_i16")
        return out

    def _preemph_inplace(self, x):
        lib = _hip.lib()
        B, L = x.shape
        nbytes = lib.air_preemph_ws_bytes(_hip.ci(B), _hip.ci(L))
        ws = torch.empty(max(1, nbytes // 4), device=x.device, dtype=torch.float32)
        _hip.check(lib.air_preemph_inplace(_hip.dptr(x), _hip.ci(B), _hip.ci(L), _hip.cf(0.97),
                                           _hip.dptr(ws), _hip.csz(nbytes), _hip.stream()),
                   "air_preemph_inplace")
