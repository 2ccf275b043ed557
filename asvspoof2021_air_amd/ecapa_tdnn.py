"""Drop-in for the reference's ``ecapa_tdnn.Res2Net2`` (ECAPA-TDNN, ecapa_tdnn.py:97-198)
with ``Bottle2neck`` (:31-95) and ``SEModule`` (:15-29).

Same constructors, ``forward(x:(B,n_mels,T)) -> (feat:(B,256), out:(B,nOut))``, the
reference's 248 ``state_dict`` keys and conv -> ReLU -> BN ordering.  Forward and backward run
in the gfx950 kernels of csrc/conv2d.hip (conv1d entry points), norm_act.hip and
ecapa_ops.hip; this file only sequences them inside one ``autograd.Function``.

Layout choices that remove the reference's copies:
  * ``torch.split`` / ``torch.cat`` of the Res2 branch and ``cat(x1,x2,x3)`` are channel-slice
    views (batch-strided kernel arguments): every block writes straight into its slice;
  * the (B,4608,T) context tensor (ecapa_tdnn.py:178) is never built: the tiled mean/std
    part of ``attention.0`` is constant over T, so it is applied as a per-utterance bias
    W[:,1536:] @ [mean; std] in the conv epilogue (same arithmetic, re-associated).
``compute_dtype`` (``model.set_compute_dtype``):
  * "fp32": the reference's arithmetic;
  * "bf16" (BASELINE.json configs[2], round 3): bf16-RESIDENT activations - every (B, C, T) tensor from the first
    BatchNorm's output to the pooling lives in HBM as bf16 rows (csrc/ecapa_bf16.hip), the K = 1 and the dilated
    K = 3 convolutions run forward, dgrad and (K = 1) wgrad on the bf16 matrix cores with fp32 accumulation and
    write bf16; BatchNorm / SE / pooling read bf16, compute in fp32 and round each stored value once.  The tensors a
    torch.autocast(bfloat16) run of the reference holds in bf16 are the same; conv1 (K = 5, fp32 input), the
    statistics, the SE and pooled vectors, fc6 and every parameter / gradient stay fp32 (oracle/ecapa.py,
    bf16="resident", states each rounding);
  * "bf16c" (rounds 1-2): bf16 COMPUTE only - the same contractions on the bf16 matrix cores, operands rounded as
    they are staged, but every tensor fp32 in HBM (plus bf16 operand copies for the weight gradients).
"""
import math

import os

import torch
import torch.nn as nn

from . import _hip, ops
from . import ops_h as oh
from .arena import ParamArena


class SEModule(nn.Module):
    """Parameter holder for ecapa_tdnn.py:15-29."""

    def __init__(self, channels, bottleneck=128):
        super().__init__()
        self.se = nn.Sequential(
            nn.AdaptiveAvgPool1d(1),
            nn.Conv1d(channels, bottleneck, kernel_size=1, padding=0),
            nn.ReLU(),
            nn.BatchNorm1d(bottleneck),
            nn.Conv1d(bottleneck, channels, kernel_size=1, padding=0),
            nn.Sigmoid(),
        )

    def forward(self, input):
        """Stand-alone forward of ecapa_tdnn.py:27-29 (fp32, forward only): x * sigmoid(se(x))."""
        _forward_only(self, input, "SEModule")
        x = input.float().contiguous()
        B, C, T = x.shape
        se = self.se
        det = lambda p: p.detach()
        m, _ = ops.row_stats(x, want_std=False)
        z1 = ops.linear_fwd(m, det(se[1].weight).view(se[1].out_channels, -1), det(se[1].bias), relu=True)
        st = _bn(z1.view(B, -1, 1), se[3], self.training)
        z1n = ops.bn_apply(z1.view(B, -1, 1), st[2], st[3]).view(B, -1)
        z2 = ops.linear_fwd(z1n, det(se[4].weight).view(se[4].out_channels, -1), det(se[4].bias))
        out = torch.empty_like(x)
        ops.se_scale_fwd(x, z2, torch.zeros_like(x), out)
        ops.bn_flush()
        return out


class Bottle2neck(nn.Module):
    """Parameter holder for ecapa_tdnn.py:31-95."""

    def __init__(self, inplanes, planes, kernel_size=None, dilation=None, scale=4):
        super().__init__()
        width = int(math.floor(planes / scale))
        self.conv1 = nn.Conv1d(inplanes, width * scale, kernel_size=1)
        self.bn1 = nn.BatchNorm1d(width * scale)
        self.nums = scale - 1
        convs, bns = [], []
        num_pad = math.floor(kernel_size / 2) * dilation
        for i in range(self.nums):
            convs.append(nn.Conv1d(width, width, kernel_size=kernel_size, dilation=dilation, padding=num_pad))
            bns.append(nn.BatchNorm1d(width))
        self.convs = nn.ModuleList(convs)
        self.bns = nn.ModuleList(bns)
        self.conv3 = nn.Conv1d(width * scale, planes, kernel_size=1)
        self.bn3 = nn.BatchNorm1d(planes)
        self.relu = nn.ReLU()
        self.width = width
        self.dilation = dilation
        self.se = SEModule(planes)

    def forward(self, x):
        """Stand-alone forward of ecapa_tdnn.py:64-95 (fp32, forward only), composed from the kernels
        ``Res2Net2.forward`` sequences: conv1 -> ReLU -> BN, the Res2 chain of dilated convs, conv3 -> ReLU -> BN,
        SE gate, + x."""
        _forward_only(self, x, "Bottle2neck")
        x = x.float().contiguous()
        B, C, T = x.shape
        w, d, nums = self.width, self.dilation, self.nums
        det = lambda p: p.detach()
        training = self.training
        r1 = ops.conv1d_fwd(x, det(self.conv1.weight), det(self.conv1.bias), relu=True)
        st1 = _bn(r1, self.bn1, training)
        o1 = ops.bn_apply(r1, st1[2], st1[3])
        cat = torch.empty_like(o1)
        t_i = o1[:, :w]
        for i in range(nums):
            r_i = ops.conv1d_fwd(t_i, det(self.convs[i].weight), det(self.convs[i].bias), relu=True, dil=d, pad=d)
            st_i = _bn(r_i, self.bns[i], training)
            if i + 1 < nums:
                t_next = torch.empty((B, w, T), device=x.device, dtype=torch.float32)
                ops.res2_bn_apply(r_i, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w], o1[:, (i + 1) * w:(i + 2) * w], t_next)
            else:
                t_next = None
                ops.res2_bn_apply(r_i, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w])
            t_i = t_next
        ops.add_strided(cat[:, nums * w:], o1[:, nums * w:])
        r3 = ops.conv1d_fwd(cat, det(self.conv3.weight), det(self.conv3.bias), relu=True)
        st3 = _bn(r3, self.bn3, training)
        o3 = ops.bn_apply(r3, st3[2], st3[3])
        se = self.se.se
        m, _ = ops.row_stats(o3, want_std=False)
        z1 = ops.linear_fwd(m, det(se[1].weight).view(se[1].out_channels, -1), det(se[1].bias), relu=True)
        stS = _bn(z1.view(B, -1, 1), se[3], training)
        z1n = ops.bn_apply(z1.view(B, -1, 1), stS[2], stS[3]).view(B, -1)
        z2 = ops.linear_fwd(z1n, det(se[4].weight).view(se[4].out_channels, -1), det(se[4].bias))
        out = torch.empty_like(x)
        ops.se_scale_fwd(o3, z2, x, out)
        ops.bn_flush()
        return out


def _forward_only(mod, x, name):
    if not x.is_cuda:
        raise _hip.AirError("%s HIP path needs a GPU tensor; there is no CPU fallback" % name)
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mod.parameters())):
        raise NotImplementedError("%s.forward is forward-only (use torch.no_grad()); gradients flow through "
                                  "Res2Net2.forward" % name)


def _bn(x3, bn, training):
    """BatchNorm1d on a (B, C, S) tensor: returns (mean, invstd, scale, shift)."""
    if training:
        st = ops.bn_stats(x3, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                          bn.eps, bn.momentum)
        ops.bn_tick(bn.num_batches_tracked)
        return st
    scale, shift = ops.bn_eval_coeffs(bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                      bn.running_var, bn.eps)
    return None, None, scale, shift


class _EcapaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        ctx.set_materialize_grads(False)
        feat, out, saved = model._forward_impl(x, save=True)
        ctx.model, ctx.saved = model, saved
        return feat, out

    @staticmethod
    def backward(ctx, dfeat, dout):
        model, saved = ctx.model, ctx.saved
        ctx.saved = None
        return (None, None) + tuple(model._backward_impl(saved, dfeat, dout))


class Res2Net2(nn.Module):
    def __init__(self, block, C, model_scale, nOut, n_mels, encoder_type="ECA", context=True,
                 summed=False, out_bn=True, **kwargs):
        self.context = context
        self.summed = summed
        self.n_mfcc = n_mels
        self.encoder_type = encoder_type
        self.out_bn = out_bn
        super().__init__()
        if encoder_type not in ("ECA", "ASP"):
            raise ValueError("Undefined encoder")  # ecapa_tdnn.py:135-136
        self.scale = model_scale
        self.conv1 = nn.Conv1d(self.n_mfcc, C, kernel_size=5, stride=1, padding=2)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(C)
        self.layer1 = block(C, C, kernel_size=3, dilation=2, scale=self.scale)
        self.layer2 = block(C, C, kernel_size=3, dilation=3, scale=self.scale)
        self.layer3 = block(C, C, kernel_size=3, dilation=4, scale=self.scale)
        self.layer4 = nn.Conv1d(3 * C, 1536, kernel_size=1)
        self.instancenorm = nn.InstanceNorm1d(self.n_mfcc)  # constructed, never applied (:120)
        # context=False / summed=True (round 6; the reference's own score files lfcc_ecapa512c{t,f}s{t,f}_* were made with
        # them): served by the fp32 path; the bf16 paths are built for the trainer's defaults (main_train.py:167)
        attn_input = 1536 * 3 if self.context else 1536  # :126-129
        # 'ASP' (:133-134): ONE attention weight per frame, shared by the 1536 channels.  Served (fp32 path) by the ECA
        # kernels on the layer's weight row repeated 1536 times - every channel then carries the same logits, which is
        # what the reference's broadcast of w (B, 1, T) over x (B, 1536, T) computes (:184-185); the gradient of the
        # one row is the sum of the 1536 rows' gradients
        attn_output = 1536 if encoder_type == "ECA" else 1
        self.attention = nn.Sequential(
            nn.Conv1d(attn_input, 128, kernel_size=1),
            nn.ReLU(),
            nn.BatchNorm1d(128),
            nn.Conv1d(128, attn_output, kernel_size=1),
            nn.Softmax(dim=2),
        )
        self.bn5 = nn.BatchNorm1d(3072)
        self.fc6 = nn.Linear(3072, 256)
        self.fc7 = nn.Linear(256, nOut)
        self.bn7 = nn.BatchNorm1d(nOut)
        self.C = C
        self._arena = None
        self.compute_dtype = "fp32"
        self._bucketer = None  # dist.GradBucketer when the all-reduce is overlapped with backward
        # weight gradients on a side HIP stream (they feed nothing until the optimiser): the MFMA-bound GEMMs and the
        # small K = 3 kernels overlap the HBM-bound BatchNorm / pooling backward passes of the main stream
        self.overlap_wgrad = os.environ.get("AIR_OVERLAP_WGRAD", "1") == "1"
        self.fuse_tap_stats = os.environ.get("AIR_TAP_STATS", "1") == "1"  # Res2 branch statistics from the conv epilogue
        self.fuse_pw_stats = os.environ.get("AIR_PW_STATS", "1") == "1"    # K = 1 convs: statistics from the GEMM epilogue
        # (round 6) Res2 chain: the elementwise pass in front of a branch conv is that conv's prologue (bf16-resident path)
        # bit 1: forward (h_res2_kernel's join), bit 2: backward (h_bn_bwd_apply_kernel).  Measured (profiles/
        # r06_res2_chain.md): forward 16.80 k utt/s against 16.79 k unfused (18 launches and 0.45 GB per step fewer, no
        # time: the conv's fixed ~8 us per launch hides nothing of the extra streams), backward 16.56 k (-1.3 %: the
        # fused launch is 29.9 us where conv 14.8 + apply 9.4 were 24.2) - so forward on, backward off
        self.fuse_tap_prologue = int(os.environ.get("AIR_TAP_PROLOGUE", "1"))
        self._side_stream = None
        # Under hipGraph capture (train.Trainer.enable_graph) a fork per weight gradient makes a graph with ~14 cross-stream
        # edges, which ROCm replays slower than one chain (round 4).  "batched" (experiment, AIR_WGRAD_BATCHED=1): the
        # bf16-resident backward queues its weight-gradient launches and hands them to the side stream at FOUR points
        # only - in front of each block's Res2 chain (21 short dependent launches that leave most of the chip idle) and
        # at the end: 4 forks and 1 join.  Measured (round 5): still 5 ms of host time per replay and a slower step than
        # one chain; left off.
        self.wgrad_batched = False

    def enable_ddp_overlap(self, bucket_bytes=8 << 20):
        """Launch the gradient all-reduce from inside backward (one process per GPU, world size > 1):
        layer4 + attention + bn5 + fc6 (15.7 of the 25 MB) leave as soon as layer4's weight gradient is
        enqueued, underneath the three Bottle2neck blocks' backward."""
        from .dist import GradBucketer
        self._bucketer = GradBucketer(bucket_bytes)
        return self

    def __getstate__(self):
        """Whole-module pickles (main_train.py:675-704): the flat arenas are rebuilt on first use."""
        st = dict(self.__dict__)
        st["_arena"] = None
        st["_bucketer"] = None
        st["_segment_cut"] = None
        st["_side_stream"] = None
        return st

    def set_compute_dtype(self, dtype):
        if dtype not in ("fp32", "bf16", "bf16c"):
            raise ValueError("compute_dtype must be 'fp32', 'bf16' or 'bf16c', got %r" % (dtype,))
        self.compute_dtype = dtype
        return self

    def forward_saved(self, x):
        """The train-mode forward WITHOUT autograd: (feat, saved) - see ResNet.forward_saved."""
        x = x.float().contiguous()
        self.arena()
        feat, _, saved = self._forward_impl(x, save=True)
        return feat, saved

    def backward_saved(self, saved, dfeat):
        return self._backward_impl(saved, dfeat, None)

    # ------------------------------------------------------------------ plumbing
    def arena(self):
        dev = self.conv1.weight.device
        if self._arena is None:
            self._arena = ParamArena(list(self.named_parameters()),
                                     tail_names=("fc7.weight", "fc7.bias", "bn7.weight", "bn7.bias"))
        if not self._arena.bound() or self._arena.device != dev:
            self._arena.bind(dev)
        return self._arena

    def forward(self, x):
        if not x.is_cuda:
            raise _hip.AirError("Res2Net2 HIP path needs a GPU tensor; there is no CPU fallback")
        if x.dim() != 3 or x.shape[1] != self.n_mfcc:
            raise ValueError("Res2Net2 expects (B, %d, T), got %s" % (self.n_mfcc, tuple(x.shape)))
        x = x.float().contiguous()
        arena = self.arena()
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for _, p, _, _ in arena.entries):
            return _EcapaFn.apply(self, x, *[p for _, p, _, _ in arena.entries])
        feat, out, _ = self._forward_impl(x, save=False)
        return feat, out

    # ------------------------------------------------------------------ forward
    def _block_fwd(self, blk, inp, out, training, save, out_bf=None):
        """Bottle2neck (ecapa_tdnn.py:64-95).  ``inp`` / ``out`` may be channel-slice views."""
        B, C, T = inp.shape
        w, d, nums = blk.width, blk.dilation, blk.nums
        det = lambda p: p.detach()
        bf = getattr(self, "_bf16c_now", self.compute_dtype == "bf16c")
        r1 = ops.conv1d_fwd(inp, det(blk.conv1.weight), det(blk.conv1.bias), relu=True, bf16=bf)
        st1 = _bn(r1, blk.bn1, training)
        o1 = ops.bn_apply(r1, st1[2], st1[3])
        cat = torch.empty_like(o1)
        # bf16 training: the concat's bf16 copy (conv3's weight-gradient operand) is written by the passes that fill it
        cat_bf = ops.bf16_rows(None, B, C, T, inp.device) if (bf and save) else None
        cb = (lambda lo, hi: cat_bf[:, lo:hi]) if cat_bf is not None else (lambda lo, hi: None)
        t_list, r_list, st_list = [], [], []
        t_i = o1[:, :w]
        # bf16: the seven branch weights (a regular stride apart in the parameter arena) are packed in one launch
        wp = ops.conv1d_tap_pack([det(c.weight) for c in blk.convs], transpose=False) if bf else None
        for i in range(nums):
            r_i = ops.conv1d_fwd(t_i, det(blk.convs[i].weight), det(blk.convs[i].bias), relu=True,
                                 dil=d, pad=d, bf16=bf, w_packed=wp[i] if wp is not None else None)
            st_i = _bn(r_i, blk.bns[i], training)
            # BN-apply, store into the concat slice and form the next branch's input in one pass
            if i + 1 < nums:
                t_next = torch.empty((B, w, T), device=inp.device, dtype=torch.float32)
                ops.res2_bn_apply(r_i, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w],
                                  o1[:, (i + 1) * w:(i + 2) * w], t_next, y1_bf=cb(i * w, (i + 1) * w))
            else:
                t_next = None
                ops.res2_bn_apply(r_i, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w], y1_bf=cb(i * w, (i + 1) * w))
            t_list.append(t_i)
            r_list.append(r_i)
            st_list.append(st_i)
            t_i = t_next
        ops.add_strided(cat[:, nums * w:], o1[:, nums * w:], out_bf=cb(nums * w, C))
        r3 = ops.conv1d_fwd(cat, det(blk.conv3.weight), det(blk.conv3.bias), relu=True, bf16=bf)
        st3 = _bn(r3, blk.bn3, training)
        se = blk.se.se
        if 32 <= T <= 1024:  # the SE squeeze (mean over time) comes out of the pass that writes o3
            m = torch.empty((B, C), device=inp.device, dtype=torch.float32)
            o3 = ops.bn_apply(r3, st3[2], st3[3], rowmean=m)
        else:
            o3 = ops.bn_apply(r3, st3[2], st3[3])
            m, _ = ops.row_stats(o3, want_std=False)
        z1 = ops.linear_fwd(m, det(se[1].weight).view(se[1].out_channels, -1), det(se[1].bias), relu=True)
        stS = _bn(z1.view(B, -1, 1), se[3], training)
        z1n = ops.bn_apply(z1.view(B, -1, 1), stS[2], stS[3]).view(B, -1)
        z2 = ops.linear_fwd(z1n, det(se[4].weight).view(se[4].out_channels, -1), det(se[4].bias))
        ops.se_scale_fwd(o3, z2, inp, out, out_bf=out_bf)
        if save:
            return dict(bf16c=bf, blk=blk, inp=inp, r1=r1, st1=st1, o1=o1, t=t_list, r=r_list, st=st_list, cat=cat, cat_bf=cat_bf,
                        r3=r3, st3=st3, o3=o3, m=m, z1=z1, stS=stS, z1n=z1n, z2=z2)
        return None

    def _forward_impl(self, x, save):
        bf = self.compute_dtype == "bf16c"
        if (not self.context or self.summed or self.encoder_type != "ECA") and self.compute_dtype != "fp32":
            raise _hip.AirError("Res2Net2(context=%s, summed=%s, encoder_type=%r): the non-default options run in "
                                "compute_dtype 'fp32' (the bf16 paths are built for main_train.py:167's defaults)" % (
                                    self.context, self.summed, self.encoder_type))
        if self.compute_dtype == "bf16":
            if oh.tp(x.shape[2]) <= oh.max_tp():
                return self._forward_h(x, save)
            #  longer than the bf16-resident kernels' register-resident rows: THIS call runs as bf16
            # compute on fp32 tensors (any length, like the reference) instead of raising; the backward follows the
            # saved tensors' kind, so nothing else has to switch
            if not getattr(self, "_warned_long", False):
                import warnings
                warnings.warn("bf16-resident ECAPA takes utterances of at most %d frames (got T = %d): this and later "
                              "longer inputs run as compute_dtype 'bf16c' (bf16 matrix cores on fp32 tensors)"
                              % (oh.max_tp(), x.shape[2]))
                self._warned_long = True
            bf = True
        self._bf16c_now = bf  # what _block_fwd sees for THIS call; the backward reads it from the saved state
        training = self.training
        det = lambda p: p.detach()
        B, _, T = x.shape
        C = self.C
        r0 = ops.conv1d_fwd(x, det(self.conv1.weight), det(self.conv1.bias), relu=True, pad=2)  # :159-160
        st0 = _bn(r0, self.bn1, training)
        # bf16 training: the first block's input and attention's hidden tensor leave their BatchNorm with a bf16 copy
        # (X operands of the conv1 / attention.3 weight gradients)
        h_bf = ops.bf16_rows(None, B, C, T, x.device) if (bf and save) else None
        h = ops.bn_apply(r0, st0[2], st0[3], y_bf=h_bf)  # :161
        cat123 = torch.empty((B, 3 * C, T), device=x.device, dtype=torch.float32)
        # bf16 training: the concat's bf16 copy [b][1536][Tp] is written slice by slice by the blocks' last kernels;
        # layer4's forward GEMM reads it K-major (no transposed copy), its weight gradient and the conv1 weight
        # gradients of blocks 2 and 3 read it in backward
        cat_bf = ops.bf16_rows(None, B, 3 * C, T, x.device) if (bf and save and T % 2 == 0) else None
        blocks = []
        inp = h
        for k, blk in enumerate((self.layer1, self.layer2, self.layer3)):
            out = cat123[:, k * C:(k + 1) * C]
            blocks.append(self._block_fwd(blk, inp, out, training, save,
                                          out_bf=cat_bf[:, k * C:(k + 1) * C] if cat_bf is not None else None))
            if self.summed:  # :163-166: the next block reads x + x1 (+ x2): the running sum, a tensor of its own
                if k < 2:
                    inp = ops.add_strided(torch.empty((B, C, T), device=x.device, dtype=torch.float32), inp, out)
            else:
                inp = out
        # bf16 training: layer4's GEMM epilogue also writes x4's bf16 copy, the X operand of attention.0's weight
        # gradient (kept until backward like x4 itself)
        x4_bf = ops.bf16_rows(None, B, self.layer4.out_channels, T, x.device) if (bf and save) else None
        x4 = None
        if cat_bf is not None:
            x4 = ops.conv1d_pointwise_kmajor(cat_bf, det(self.layer4.weight), T, bias=det(self.layer4.bias), relu=True,
                                             y_bf=x4_bf)
        if x4 is None:
            x4 = ops.conv1d_fwd(cat123, det(self.layer4.weight), det(self.layer4.bias), relu=True, bf16=bf,
                                y_bf=x4_bf)  # :172-173
        mean, std = ops.row_stats(x4, True, 1e-4)  # context statistics (:178; context=False: only the backward's
        # fused ReLU-mask pass reads them, with zero gradients)
        a0, a3 = self.attention[0], self.attention[3]
        w0 = det(a0.weight).view(128, -1)  # (128, 4608), or (128, 1536) without the context rows
        if self.context:
            ctx = torch.cat((mean, std), 1)  # plumbing: 2 x (B,1536) copies
            w_x = ops.add_strided(torch.empty((128, 1, 1536), device=x.device),
                                  w0[:, :1536].unsqueeze(1)).view(128, 1536, 1)
            w_c = ops.add_strided(torch.empty((128, 1, 3072), device=x.device),
                                  w0[:, 1536:].unsqueeze(1)).view(128, 3072)
            ctxb = ops.linear_fwd(ctx, w_c, None)  # (B,128): W[:,1536:] @ [mean; std]
        else:  # :179-180: global_x = x
            ctx, w_c, ctxb = None, None, None
            w_x = det(a0.weight)
        a1 = ops.conv1d_fwd(x4, w_x, det(a0.bias), bias_bc=ctxb, relu=True, bf16=bf)  # attention.0 + ReLU
        stA = _bn(a1, self.attention[2], training)
        a1n_bf = ops.bf16_rows(None, B, a1.shape[1], T, x.device) if (bf and save) else None
        a1n = ops.bn_apply(a1, stA[2], stA[3], y_bf=a1n_bf)
        w3, b3 = det(a3.weight), det(a3.bias)
        if self.encoder_type == "ASP":  # the one row for every channel (weight-sized plumbing copies)
            w3 = w3.expand(x4.shape[1], -1, -1).contiguous()
            b3 = b3.expand(x4.shape[1]).contiguous()
        wts = ops.conv1d_fwd(a1n, w3, b3, bf16=bf)  # logits -> softmax weights below
        pooled = ops.asp_fwd(x4, wts)  # :184-187 (mu | sg)
        st5 = _bn(pooled.view(B, -1, 1), self.bn5, training)
        p5 = ops.bn_apply(pooled.view(B, -1, 1), st5[2], st5[3]).view(B, -1)
        feat = ops.linear_fwd(p5, det(self.fc6.weight), det(self.fc6.bias))  # :191
        o7 = ops.linear_fwd(feat, det(self.fc7.weight), det(self.fc7.bias))  # :193
        st7 = None
        out = o7
        if self.out_bn:
            st7 = _bn(o7.view(B, -1, 1), self.bn7, training)
            out = ops.bn_apply(o7.view(B, -1, 1), st7[2], st7[3]).view(B, -1)
        S = None
        if save:
            if not training:
                raise NotImplementedError("backward through eval-mode BatchNorm is not on the hot path")
            S = dict(bf16c=bf, x=x, r0=r0, st0=st0, h=h, h_bf=h_bf, a1n_bf=a1n_bf, cat123=cat123, cat_bf=cat_bf, blocks=blocks, x4=x4, x4_bf=x4_bf, mean=mean,
                     std=std,
                     ctx=ctx, w_x=w_x, w_c=w_c, a1=a1, stA=stA, a1n=a1n, wts=wts, w3=w3, pooled=pooled, st5=st5,
                     p5=p5, feat=feat, o7=o7, st7=st7)
        ops.bn_flush()
        return feat, out, S

    # ----------------------------------------------------------------- backward
    def _block_bwd(self, S, dout, G, pre, inp_bf=None, add2=None, on_side=None):
        """dout: gradient w.r.t. the block output ((B,C,T), dense or a channel slice).  Returns d(inp) (+ add2) dense.
        inp_bf: bf16 copy of the block input when the caller holds one (a slice of the concat's copy).
        add2: a second tensor to fold into the returned gradient (bf16 path: the same epilogue).
        on_side(fn, *reads): runs a weight-gradient launch sequence (on the side stream when overlap is on)."""
        if on_side is None:
            on_side = lambda fn, *reads: fn()
        blk = S["blk"]
        det = lambda p: p.detach()
        bf = S.get("bf16c", self.compute_dtype == "bf16c")
        B, C, T = S["o3"].shape
        w, d, nums = blk.width, blk.dilation, blk.nums
        se = blk.se.se
        gv = lambda n: G[pre + n]
        do3, dz2 = ops.se_scale_bwd(S["o3"], S["z2"], dout)
        dz1n, _, _ = ops.linear_bwd(S["z1n"], det(se[4].weight).view(se[4].out_channels, -1), dz2, True,
                                    dw=gv("se.se.4.weight").view(se[4].out_channels, -1),
                                    db=gv("se.se.4.bias"))
        stS = S["stS"]
        dz1, _, _ = ops.bn_bwd(S["z1"].view(B, -1, 1), dz1n.view(B, -1, 1), stS[0], stS[1],
                               det(se[3].weight), det(se[3].bias), relu_in=True,
                               dgamma=gv("se.se.3.weight"), dbeta=gv("se.se.3.bias"))
        dm, _, _ = ops.linear_bwd(S["m"], det(se[1].weight).view(se[1].out_channels, -1), dz1.view(B, -1), True,
                                  dw=gv("se.se.1.weight").view(se[1].out_channels, -1),
                                  db=gv("se.se.1.bias"))
        # the SE squeeze's gradient (d mean_T / d o3 = 1/T, ecapa_tdnn.py:19) enters bn3's backward as a
        # per-(b, c) constant on the incoming gradient; the conv bias gradient comes out of the same pass
        st3 = S["st3"]
        # bf16 path: the BatchNorm backward writes the weight-gradient GEMM's dY operand itself (one buffer for
        # every 512-channel gradient of the model: same stream, so each copy is consumed before the next is made)
        dy_bf = ops.bf16_rows(pre + "dc3", B, C, T, do3.device) if bf else None
        dc3, _, _ = ops.bn_bwd(S["r3"], do3, st3[0], st3[1], det(blk.bn3.weight), det(blk.bn3.bias),
                               relu_in=True, dx=do3, dgamma=gv("bn3.weight"), dbeta=gv("bn3.bias"),
                               rowbias=dm, rowbias_scale=1.0 / T, dbias=gv("conv3.bias"), dx_bf16=dy_bf)
        on_side(lambda: ops.conv1d_wgrad(S["cat"], dc3, blk.conv3.weight.shape, out=gv("conv3.weight"), bf16=bf,
                                         dy_bf=dy_bf, x_bf=S["cat_bf"]), dc3)
        dcat = ops.conv1d_dgrad(dc3, det(blk.conv3.weight), bf16=bf)
        do1 = torch.empty_like(dcat)
        ops.add_strided(do1[:, nums * w:], dcat[:, nums * w:])
        din_next = None
        wpt = ops.conv1d_tap_pack([det(c.weight) for c in blk.convs], transpose=True) if bf else None
        for i in reversed(range(nums)):
            # d(sp_i) = d(cat slice i) + d(input of branch i + 1): both are channel-slice views and the sum is
            # formed inside the BatchNorm backward passes
            st_i = S["st"][i]
            dc_i = torch.empty((B, w, T), device=dcat.device, dtype=torch.float32)
            ops.bn_bwd(S["r"][i], dcat[:, i * w:(i + 1) * w], st_i[0], st_i[1], det(blk.bns[i].weight),
                       det(blk.bns[i].bias), relu_in=True, dx=dc_i, dy2=din_next,
                       dgamma=gv("bns.%d.weight" % i), dbeta=gv("bns.%d.bias" % i),
                       dbias=gv("convs.%d.bias" % i))
            on_side(lambda dc_i=dc_i, i=i: ops.conv1d_wgrad(S["t"][i], dc_i, blk.convs[i].weight.shape, d, d,
                                                            out=gv("convs.%d.weight" % i)), dc_i)
            # the input gradient lands in its slice of d(o1); branch i - 1 reads it from there
            din = ops.conv1d_dgrad(dc_i, det(blk.convs[i].weight), d, d, out=do1[:, i * w:(i + 1) * w], bf16=bf,
                                   w_packed=wpt[i] if wpt is not None else None)
            din_next = din if i > 0 else None
        st1 = S["st1"]
        dy_bf = ops.bf16_rows(pre + "dc1", B, C, T, do3.device) if bf else None
        dc1, _, _ = ops.bn_bwd(S["r1"], do1, st1[0], st1[1], det(blk.bn1.weight), det(blk.bn1.bias),
                               relu_in=True, dx=do1, dgamma=gv("bn1.weight"), dbeta=gv("bn1.bias"),
                               dbias=gv("conv1.bias"), dx_bf16=dy_bf)
        on_side(lambda: ops.conv1d_wgrad(S["inp"], dc1, blk.conv1.weight.shape, out=gv("conv1.weight"), bf16=bf,
                                         dy_bf=dy_bf, x_bf=inp_bf), dc1)
        # + dout: the residual branch (ecapa_tdnn.py:93), added in the dgrad epilogue
        return ops.conv1d_dgrad(dc1, det(blk.conv1.weight), accumulate=dout, bf16=bf, accumulate2=add2)

    def _backward_impl(self, S, dfeat, dout):
        if S.get("resident"):
            return self._backward_h(S, dfeat, dout)
        arena = self.arena()
        G = arena.grad_views()
        # gradient accumulation (a second backward without zero_grad): p.grad already IS the arena view, so
        # autograd's "p.grad += returned view" would double the NEW gradient instead of adding the old one.
        # Keep the old sums aside, fold them back in at the end and return None for the aliased entries
        # (same protocol as ResNet._backward_impl).
        accumulating = any(p.grad is not None and p.grad.data_ptr() == G[n].data_ptr()
                           for n, p, _, _ in arena.entries)
        old = arena.grad.clone() if accumulating else None
        det = lambda p: p.detach()
        bf = S.get("bf16c", self.compute_dtype == "bf16c")
        B, _, T = S["x"].shape
        C = self.C
        tail = ("fc7.weight", "fc7.bias", "bn7.weight", "bn7.bias")
        have_tail = dout is not None
        # Weight gradients on the side stream (see __init__).  Ordering: a weight gradient starts after the event that
        # marks its operands ready; tensors it reads are kept alive until the join (the caching allocator would hand
        # their memory back to the main stream); every bf16 operand copy has its own buffer; the main stream joins
        # the side stream before the gradients are used.
        main = torch.cuda.current_stream()
        use_side = self.overlap_wgrad and not accumulating
        if use_side and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=main.device)
        side = self._side_stream if use_side else main
        keep = []

        def on_side(fn, *reads):
            if not use_side:
                fn()
                return
            keep.extend(reads)
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                fn()
        if dfeat is None:
            dfeat = torch.zeros_like(S["feat"])
        dfeat = dfeat.contiguous()
        if have_tail:  # CE branch through fc7/bn7 (dead under ang_iso, main_train.py:355 -> 376)
            do7 = dout.contiguous()
            if self.out_bn:
                st7 = S["st7"]
                do7, _, _ = ops.bn_bwd(S["o7"].view(B, -1, 1), do7.view(B, -1, 1), st7[0], st7[1],
                                       det(self.bn7.weight), det(self.bn7.bias),
                                       dgamma=G["bn7.weight"], dbeta=G["bn7.bias"])
                do7 = do7.view(B, -1)
            dx7, _, _ = ops.linear_bwd(S["feat"], det(self.fc7.weight), do7, True, dw=G["fc7.weight"],
                                       db=G["fc7.bias"])
            dfeat = ops.add_(dx7, dfeat)
        dp5, _, _ = ops.linear_bwd(S["p5"], det(self.fc6.weight), dfeat, True, dw=G["fc6.weight"],
                                   db=G["fc6.bias"])
        st5 = S["st5"]
        dpooled, _, _ = ops.bn_bwd(S["pooled"].view(B, -1, 1), dp5.view(B, -1, 1), st5[0], st5[1],
                                   det(self.bn5.weight), det(self.bn5.bias),
                                   dgamma=G["bn5.weight"], dbeta=G["bn5.bias"])
        x4, wts = S["x4"], S["wts"]
        dx4 = torch.empty_like(x4)
        rows3 = torch.empty((B, x4.shape[1]), device=x4.device, dtype=torch.float32)
        wide_bf = ops.bf16_rows("ecapa.dlogits", B, x4.shape[1], T, x4.device) if bf else None
        ops.asp_bwd(x4, wts, S["pooled"], dpooled.view(B, -1), dx4, accumulate=False, rowsum=rows3,
                    dlogits_bf16=wide_bf)  # wts -> dlogits
        a0, a3 = self.attention[0], self.attention[3]
        asp = self.encoder_type == "ASP"
        if asp:  # the one row's gradients = the sums over the 1536 repeated rows'
            ops.sum_rows(ops.sum_rows(rows3).view(-1, 1), out=G["attention.3.bias"])

            def att3_wgrad():
                dw = ops.conv1d_wgrad(S["a1n"], wts, tuple(S["w3"].shape))
                ops.sum_rows(dw.view(dw.shape[0], -1), out=G["attention.3.weight"].view(-1))

            on_side(att3_wgrad, wts)
        else:
            ops.sum_rows(rows3, out=G["attention.3.bias"])  # analytically zero (softmax over T): rounding noise
            on_side(lambda: ops.conv1d_wgrad(S["a1n"], wts, a3.weight.shape, out=G["attention.3.weight"], bf16=bf,
                                             dy_bf=wide_bf, x_bf=S["a1n_bf"]), wts)
        da1n = ops.conv1d_dgrad(wts, S["w3"], bf16=bf)
        stA = S["stA"]
        da1_bf = ops.bf16_rows("ecapa.da1", B, 128, T, x4.device) if bf else None
        da1, _, _ = ops.bn_bwd(S["a1"], da1n, stA[0], stA[1], det(self.attention[2].weight),
                               det(self.attention[2].bias), relu_in=True, dx=da1n,
                               dgamma=G["attention.2.weight"], dbeta=G["attention.2.bias"],
                               dbias=G["attention.0.bias"], dx_bf16=da1_bf)
        gw0 = G["attention.0.weight"].view(128, -1)  # (128, 4608)

        def att0_wgrad():
            if S["ctx"] is None:  # context=False: the layer's whole weight
                ops.conv1d_wgrad(x4, da1, (128, 1536, 1), out=G["attention.0.weight"], bf16=bf, dy_bf=da1_bf, x_bf=S["x4_bf"])
                return
            dwx = ops.conv1d_wgrad(x4, da1, (128, 1536, 1), bf16=bf, dy_bf=da1_bf, x_bf=S["x4_bf"])
            ops.add_strided(gw0[:, :1536].unsqueeze(1), dwx.view(128, 1, 1536))

        on_side(att0_wgrad, da1)
        ops.conv1d_dgrad(da1, S["w_x"], accumulate=dx4, out=dx4, bf16=bf)
        if S["ctx"] is not None:
            dctxb = ops.row_sum(da1)  # (B,128)
            dctx, dwc, _ = ops.linear_bwd(S["ctx"], S["w_c"], dctxb, True, need_db=False)
            ops.add_strided(gw0[:, 1536:].unsqueeze(1), dwc.view(128, 1, 3072))
            dmean = dctx[:, :1536].contiguous()
            dstd = dctx[:, 1536:].contiguous()
        else:  # no statistics rows in the attention input: nothing flows into mean / std
            dmean = torch.zeros((B, x4.shape[1]), device=x4.device, dtype=torch.float32)
            dstd = torch.zeros_like(dmean)
        # context-statistics gradient, the ReLU after layer4 (:173) and the per-row sums for the bias
        # gradient in ONE pass over the (B, 1536, T) tensor
        rows = torch.empty((B, x4.shape[1]), device=x4.device, dtype=torch.float32)
        dx4_bf = ops.bf16_rows("ecapa.dx4", B, x4.shape[1], T, x4.device) if bf else None
        ops.row_stats_bwd(x4, S["mean"], S["std"], dmean, dstd, dx4, accumulate=True, relu_mask=True, rowsum=rows,
                          dx_bf16=dx4_bf)
        ops.sum_rows(rows, out=G["layer4.bias"])
        # the concat's bf16 copy is made once: layer4's weight gradient reads all of it, the conv1 weight gradients
        # of blocks 2 and 3 read the channel slices that were their inputs
        cat_bf = S["cat_bf"]
        made_here = bf and cat_bf is None
        if made_here:
            cat_bf = ops.bf16_rows("ecapa.cat123", B, 3 * C, T, dx4.device)

        def layer4_wgrad():
            if made_here:
                ops.conv1d_cvt_bf16(S["cat123"], cat_bf)
            ops.conv1d_wgrad(S["cat123"], dx4, self.layer4.weight.shape, out=G["layer4.weight"], bf16=bf, x_bf=cat_bf,
                             dy_bf=dx4_bf)

        on_side(layer4_wgrad, dx4)
        # data parallel: everything from layer4.weight to the end of the gradient arena is final
        bucketer = None if accumulating else getattr(self, "_bucketer", None)
        offsets = {n: o for n, _, o, _ in arena.entries}

        def grads_final_from(first_param):
            cut = getattr(self, "_segment_cut", None)
            if cut is not None:  # train.Trainer's segmented hipGraph capture: a segment may end here
                cut(offsets[first_param])
            if bucketer is not None:
                evs = [torch.cuda.Event()]
                evs[0].record(main)
                if use_side:
                    evs.append(torch.cuda.Event())
                    evs[1].record(side)
                bucketer.ready(offsets[first_param], evs)

        if bucketer is not None:
            bucketer.reset(arena.grad, arena.head_total)
        grads_final_from("layer4.weight")
        # layer4's data gradient reads d(x4)'s bf16 copy (written by the context-statistics backward) K-major
        dcat123 = ops.conv1d_pointwise_kmajor(dx4_bf, det(self.layer4.weight), T, dgrad=True) if dx4_bf is not None else None
        if dcat123 is None:
            dcat123 = ops.conv1d_dgrad(dx4, det(self.layer4.weight), bf16=bf)
        dnext = None
        # the two-operand dgrad epilogue: the fused bf16 pointwise kernels only (8-byte aligned rows; layers of
        # 1 M weights and more route to the wide GEMM, which takes a single dense accumulate operand)
        fold = bf and T % 2 == 0 and C % 128 == 0 and C * C < (1 << 20)
        dsum = None  # summed=True: the gradient of the running sum x + x1 (+ x2) = the sum of the later blocks' d(input)
        for k in (2, 1, 0):
            if fold:
                # d(block k output) = its slice of the concat gradient + d(block k + 1 input): block k + 1's last
                # dgrad already added this block's slice (add2), block 3 reads its slice in place
                dblk = dcat123[:, k * C:(k + 1) * C] if dnext is None else dnext
                add2 = dcat123[:, (k - 1) * C:k * C] if k > 0 else None
            else:
                dblk = torch.empty((B, C, T), device=dx4.device, dtype=torch.float32)
                # summed (:163-166): x_k also feeds EVERY later block's input, not only the next one's
                ops.add_strided(dblk, dcat123[:, k * C:(k + 1) * C], dsum if self.summed else dnext)
                add2 = None
            dnext = self._block_bwd(S["blocks"][k], dblk, G, "layer%d." % (k + 1),
                                    inp_bf=(cat_bf[:, (k - 1) * C:k * C] if k > 0 else S["h_bf"]) if bf else None, add2=add2,
                                    on_side=on_side)
            if self.summed:
                dsum = dnext if dsum is None else ops.add_(dsum, dnext)
            grads_final_from("layer%d.conv1.weight" % (k + 1))
        if self.summed:
            dnext = dsum  # d(h) = d(input of block 1) + d(input of block 2) + d(input of block 3)
        st0 = S["st0"]
        dc0, _, _ = ops.bn_bwd(S["r0"], dnext, st0[0], st0[1], det(self.bn1.weight), det(self.bn1.bias),
                               relu_in=True, dx=dnext, dgamma=G["bn1.weight"], dbeta=G["bn1.bias"],
                               dbias=G["conv1.bias"])
        ops.conv1d_wgrad(S["x"], dc0, self.conv1.weight.shape, 1, 2, out=G["conv1.weight"])
        if use_side:
            main.wait_stream(side)  # join: every weight gradient is in the arena
        del keep[:]
        arena.tail_has_grad = have_tail
        if accumulating:
            if not have_tail:  # the tail got no new gradient: its old sums must survive the add below unchanged
                arena.grad[arena.head_total:].zero_()
            ops.add_(arena.grad, old)
            arena.tail_has_grad = True  # old tail sums may be live; the optimiser covers the whole arena
            return [None if (p.grad is not None and p.grad.data_ptr() == G[n].data_ptr())
                    else (G[n] if (have_tail or n not in tail) else None) for n, p, _, _ in arena.entries]
        return [G[n] if (have_tail or n not in tail) else None for n, _, _, _ in arena.entries]

    # =================================================================== bf16-resident path (compute_dtype "bf16")
    def _conv1_rows(self):
        """Rows of the unfolded first-layer input: Cin * K padded to the weight-gradient GEMM's 128-row tiles."""
        n = self.conv1.in_channels * self.conv1.kernel_size[0]
        return (n + 127) // 128 * 128

    def _conv1_matrix(self):
        """conv1.weight (C, Cin, K) as the (C, rows, 1) matrix of the unfolded GEMM (zero columns behind Cin * K)."""
        R0, nk = self._conv1_rows(), self.conv1.in_channels * self.conv1.kernel_size[0]
        w = self.conv1.weight.detach()
        m = getattr(self, "_conv1_wm", None)
        if m is None or m.device != w.device or m.shape[1] != R0:
            m = torch.zeros((self.C, R0, 1), device=w.device, dtype=torch.float32)
            self._conv1_wm = m
        ops.add_strided(m.view(self.C, 1, R0)[:, :, :nk], w.view(self.C, 1, nk))
        return m

    def _bn_h(self, x, T, bn, training, stats_in=None):
        """BatchNorm1d on resident rows: (mean, invstd, scale, shift).  stats_in: statistics records of x from the
        epilogue of the convolution that produced it."""
        if training:
            st = oh.bn_stats(x, T, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                             stats_in=stats_in)
            ops.bn_tick(bn.num_batches_tracked)
            return st
        scale, shift = ops.bn_eval_coeffs(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        return None, None, scale, shift

    def _pw_bn_h(self, x, w, T, bias, bn, training, bias_bc=None):
        """conv (K = 1) -> ReLU -> BatchNorm statistics on resident rows: (r, st); in training the statistics leave the
        GEMM's epilogue (round 4) instead of a pass over r."""
        if training and getattr(self, "fuse_pw_stats", True):
            r, rec = oh.conv_pointwise(x, w, T, bias=bias, bias_bc=bias_bc, relu=True, stats=True)
            return r, self._bn_h(r, T, bn, training, stats_in=rec)
        r = oh.conv_pointwise(x, w, T, bias=bias, bias_bc=bias_bc, relu=True)
        return r, self._bn_h(r, T, bn, training)

    def _block_fwd_h(self, blk, inp, out, T, training, save):
        """Bottle2neck (ecapa_tdnn.py:64-95) on resident rows.  ``inp`` / ``out`` may be channel-slice views.
        o1 = bn1(relu(conv1(inp))) is written into the buffer that becomes the concat: branch i's BatchNorm output
        replaces o1's slice i after that slice has been consumed, the pass-through group (:85) never moves."""
        B, C, Tp = inp.shape
        w, d, nums = blk.width, blk.dilation, blk.nums
        det = lambda p: p.detach()
        dev = inp.device
        r1, st1 = self._pw_bn_h(inp, det(blk.conv1.weight), T, det(blk.conv1.bias), blk.bn1, training)
        cat = oh.bn_apply(r1, T, st1[2], st1[3])
        t_i = oh.copy(cat[:, :w], oh.rows(B, w, T, dev))  # branch 0's input outlives its slice (weight gradient)
        wp = ops.conv1d_tap_pack([det(c.weight) for c in blk.convs], transpose=False)
        t_list, r_list, st_list = [], [], []
        # (round 6) the BatchNorm-apply + join of branch i - 1 (h_res2_kernel: y1 -> its concat slice, t_i = y1 + o1's
        # slice i) is the PROLOGUE of branch i's conv: that launch reads r_{i-1} and the slice, computes its operand while
        # staging it and writes y1 and t_i as side outputs - same arithmetic, same rounding points, 6 launches fewer per
        # block.  Only the last branch's BatchNorm output still takes a pass of its own.
        fusep = bool(int(getattr(self, "fuse_tap_prologue", 1)) & 1) and oh.tap_pro_ok(w, w)
        for i in range(nums):
            # (round 4: the branch's BatchNorm statistics leave the conv's epilogue - no pass over r_i)
            fuse = training and getattr(self, "fuse_tap_stats", True)
            pro, x_i = None, t_i
            if fusep and i > 0:
                t_i = oh.rows(B, w, T, dev)
                pro = oh.res2_prologue(st_list[i - 1][2], st_list[i - 1][3], add=cat[:, i * w:(i + 1) * w],
                                       y1=cat[:, (i - 1) * w:i * w], t_out=t_i)
                x_i = r_list[i - 1]
            r_i, rec_i = oh.conv_tap(x_i, wp[i], T, d, w, w, bias=det(blk.convs[i].bias), relu=True, stats=True, pro=pro) if fuse \
                else (oh.conv_tap(x_i, wp[i], T, d, w, w, bias=det(blk.convs[i].bias), relu=True, pro=pro), None)
            st_i = self._bn_h(r_i, T, blk.bns[i], training, stats_in=rec_i)
            t_next = None
            if fusep:
                if i + 1 == nums:
                    oh.res2_bn_apply(r_i, T, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w])
            elif i + 1 < nums:
                t_next = oh.rows(B, w, T, dev)
                oh.res2_bn_apply(r_i, T, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w],
                                 add=cat[:, (i + 1) * w:(i + 2) * w], y2=t_next)
            else:
                oh.res2_bn_apply(r_i, T, st_i[2], st_i[3], cat[:, i * w:(i + 1) * w])
            t_list.append(t_i)
            r_list.append(r_i)
            st_list.append(st_i)
            t_i = t_next
        r3, st3 = self._pw_bn_h(cat, det(blk.conv3.weight), T, det(blk.conv3.bias), blk.bn3, training)
        m = torch.empty((B, C), device=dev, dtype=torch.float32)
        o3 = oh.bn_apply(r3, T, st3[2], st3[3], rowmean=m)  # + the SE squeeze of the stored tensor
        se = blk.se.se
        z1 = ops.linear_fwd(m, det(se[1].weight).view(se[1].out_channels, -1), det(se[1].bias), relu=True)
        stS = _bn(z1.view(B, -1, 1), se[3], training)
        z1n = ops.bn_apply(z1.view(B, -1, 1), stS[2], stS[3]).view(B, -1)
        z2 = ops.linear_fwd(z1n, det(se[4].weight).view(se[4].out_channels, -1), det(se[4].bias))
        oh.se_scale_fwd(o3, z2, inp, T, out)
        if save:
            return dict(blk=blk, inp=inp, r1=r1, st1=st1, t=t_list, r=r_list, st=st_list, cat=cat, r3=r3, st3=st3,
                        o3=o3, m=m, z1=z1, stS=stS, z1n=z1n, z2=z2)
        return None

    def _forward_h(self, x, save):
        if self.C % 64 != 0 or self.layer1.width % 64 != 0:
            raise _hip.AirError("bf16-resident ECAPA needs C %% 64 == 0 and Res2 branches of width %% 64 == 0 (C = %d, "
                                "width %d): use compute_dtype 'bf16c' or 'fp32' for this configuration" % (
                                    self.C, self.layer1.width))
        training = self.training
        det = lambda p: p.detach()
        B, _, T = x.shape
        if oh.tp(T) > oh.max_tp():
            # the wave-per-row statistics / pooling kernels of csrc/ecapa_bf16.hip keep a whole row in registers
            raise _hip.AirError("bf16-resident ECAPA takes utterances of at most %d frames (got T = %d): use "
                                "set_compute_dtype('bf16c') or 'fp32' for longer inputs (any length, like the "
                                "reference)" % (oh.max_tp(), T))
        C = self.C
        dev = x.device
        # conv1 (K = 5 on the fp32 features, :159-161) as a pointwise GEMM on the input unfolded into bf16 rows (autocast
        # runs this layer in bf16 as well): its ReLU output and its BatchNorm output are resident tensors like the rest
        xcol = oh.unfold(x.contiguous(), self.conv1.kernel_size[0], 1, self.conv1.padding[0], self._conv1_rows())
        r0, st0 = self._pw_bn_h(xcol, self._conv1_matrix(), T, det(self.conv1.bias), self.bn1, training)
        h = oh.bn_apply(r0, T, st0[2], st0[3])
        cat123 = oh.rows(B, 3 * C, T, dev)
        blocks = []
        inp = h
        for k, blk in enumerate((self.layer1, self.layer2, self.layer3)):
            out = cat123[:, k * C:(k + 1) * C]
            blocks.append(self._block_fwd_h(blk, inp, out, T, training, save))
            inp = out
        x4 = oh.conv_pointwise(cat123, det(self.layer4.weight), T, bias=det(self.layer4.bias), relu=True)  # :172-173
        mean, std = oh.row_stats(x4, T, True, 1e-4)  # context statistics (:178)
        ctx = torch.cat((mean, std), 1)
        a0, a3 = self.attention[0], self.attention[3]
        w0 = det(a0.weight).view(128, -1)
        w_x = ops.add_strided(torch.empty((128, 1, 1536), device=dev), w0[:, :1536].unsqueeze(1)).view(128, 1536, 1)
        w_c = ops.add_strided(torch.empty((128, 1, 3072), device=dev), w0[:, 1536:].unsqueeze(1)).view(128, 3072)
        ctxb = ops.linear_fwd(ctx, w_c, None)  # (B,128): W[:,1536:] @ [mean; std], a per-utterance bias
        a1, stA = self._pw_bn_h(x4, w_x, T, det(a0.bias), self.attention[2], training, bias_bc=ctxb)  # attention.0 + ReLU
        a1n = oh.bn_apply(a1, T, stA[2], stA[3])
        wts = oh.conv_pointwise(a1n, det(a3.weight), T, bias=det(a3.bias))  # logits -> softmax weights below
        pooled = oh.asp_fwd(x4, wts, T)  # :184-187 (mu | sg)
        st5 = _bn(pooled.view(B, -1, 1), self.bn5, training)
        p5 = ops.bn_apply(pooled.view(B, -1, 1), st5[2], st5[3]).view(B, -1)
        feat = ops.linear_fwd(p5, det(self.fc6.weight), det(self.fc6.bias))  # :191
        o7 = ops.linear_fwd(feat, det(self.fc7.weight), det(self.fc7.bias))  # :193
        st7 = None
        out = o7
        if self.out_bn:
            st7 = _bn(o7.view(B, -1, 1), self.bn7, training)
            out = ops.bn_apply(o7.view(B, -1, 1), st7[2], st7[3]).view(B, -1)
        S = None
        if save:
            if not training:
                raise NotImplementedError("backward through eval-mode BatchNorm is not on the hot path")
            S = dict(resident=True, T=T, x=x, xcol=xcol, r0=r0, st0=st0, h=h, cat123=cat123, blocks=blocks, x4=x4, mean=mean, std=std,
                     ctx=ctx, w_x=w_x, w_c=w_c, a1=a1, stA=stA, a1n=a1n, wts=wts, pooled=pooled, st5=st5, p5=p5,
                     feat=feat, o7=o7, st7=st7)
        ops.bn_flush()
        return feat, out, S

    def _block_bwd_h(self, S, dout, T, G, pre, add2, on_side, flush_side=None):
        """dout: gradient w.r.t. the block output (resident rows or a channel slice).  Returns d(inp) + dout + add2.
        flush_side (batched weight gradients): called in front of the Res2 chain - what has been queued runs beside it."""
        blk = S["blk"]
        det = lambda p: p.detach()
        B, C, Tp = S["o3"].shape
        w, d, nums = blk.width, blk.dilation, blk.nums
        se = blk.se.se
        gv = lambda n: G[pre + n]
        do3, dz2 = oh.se_scale_bwd(S["o3"], S["z2"], dout, T)
        dz1n, _, _ = ops.linear_bwd(S["z1n"], det(se[4].weight).view(se[4].out_channels, -1), dz2, True,
                                    dw=gv("se.se.4.weight").view(se[4].out_channels, -1), db=gv("se.se.4.bias"))
        stS = S["stS"]
        dz1, _, _ = ops.bn_bwd(S["z1"].view(B, -1, 1), dz1n.view(B, -1, 1), stS[0], stS[1], det(se[3].weight),
                               det(se[3].bias), relu_in=True, dgamma=gv("se.se.3.weight"), dbeta=gv("se.se.3.bias"))
        dm, _, _ = ops.linear_bwd(S["m"], det(se[1].weight).view(se[1].out_channels, -1), dz1.view(B, -1), True,
                                  dw=gv("se.se.1.weight").view(se[1].out_channels, -1), db=gv("se.se.1.bias"))
        st3 = S["st3"]
        dc3 = oh.bn_bwd(S["r3"], do3, T, st3[0], st3[1], det(blk.bn3.weight), gv("bn3.weight"), gv("bn3.bias"), dx=do3,
                        rowbias=dm, rowbias_scale=1.0 / T, dbias=gv("conv3.bias"))
        on_side(lambda: oh.conv_wgrad(S["cat"], dc3, T, gv("conv3.weight")), dc3)
        dcat = oh.conv_pointwise(dc3, det(blk.conv3.weight), T, dgrad=True)  # becomes d(o1) slice by slice
        wpt = ops.conv1d_tap_pack([det(c.weight) for c in blk.convs], transpose=True)
        din_next = sums_next = None
        dcs = [None] * nums
        fuse = getattr(self, "fuse_tap_stats", True)
        if flush_side is not None:
            flush_side()
        # (round 6) the BatchNorm-backward APPLY of branch i (h_bn_bwd_apply_kernel: 21 passes per step) is the prologue
        # of that branch's data-gradient conv, which reads r_i + the two gradient halves, computes dc_i while staging it
        # and writes it out for the weight gradient.  Its output (d t_i) goes into a second tensor, do1, slice by slice
        # (the launch reads dcat's slice with a halo, so it cannot overwrite it in place); do1 = d(o1) at the end.
        fusep = bool(int(getattr(self, "fuse_tap_prologue", 1)) & 2) and oh.tap_pro_ok(w, w) and fuse
        if fusep:
            do1 = oh.rows(B, C, T, dcat.device)
            oh.copy(dcat[:, nums * w:], do1[:, nums * w:])  # the pass-through group (ecapa_tdnn.py:85)
        for i in reversed(range(nums)) if fusep else ():
            st_i = S["st"][i]
            dy_i = dcat[:, i * w:(i + 1) * w]
            oh.bn_bwd(S["r"][i], dy_i, T, st_i[0], st_i[1], det(blk.bns[i].weight), gv("bns.%d.weight" % i),
                      gv("bns.%d.bias" % i), dy2=din_next, dbias=gv("convs.%d.bias" % i), sums_in=sums_next, apply=False)
            dcs[i] = oh.rows(B, w, T, dcat.device)
            pro = oh.bn_bwd_prologue(dy_i, din_next, st_i[0], st_i[1], det(blk.bns[i].weight), gv("bns.%d.weight" % i),
                                     gv("bns.%d.bias" % i), dcs[i])
            out_i = do1[:, i * w:(i + 1) * w]
            if i > 0:
                st_p = S["st"][i - 1]
                din_next, sums_next = oh.conv_tap(S["r"][i], wpt[i], T, d, w, w, dgrad=True, out=out_i, pro=pro,
                                                  bn=(S["r"][i - 1], dcat[:, (i - 1) * w:i * w], st_p[0], st_p[1]))
            else:
                oh.conv_tap(S["r"][i], wpt[i], T, d, w, w, dgrad=True, out=out_i, pro=pro)
        if fusep:
            dcat = do1
        for i in reversed(range(nums)) if not fusep else ():
            st_i = S["st"][i]
            dc_i = oh.bn_bwd(S["r"][i], dcat[:, i * w:(i + 1) * w], T, st_i[0], st_i[1], det(blk.bns[i].weight),
                             gv("bns.%d.weight" % i), gv("bns.%d.bias" % i), dy2=din_next, dbias=gv("convs.%d.bias" % i),
                             sums_in=sums_next)
            dcs[i] = dc_i
            if i > 0 and fuse:
                # (round 4) the data gradient that joins branch i - 1's slice of the concat gradient on its way into that
                # branch's BatchNorm: the sums of that BatchNorm's backward leave this launch's epilogue
                st_p = S["st"][i - 1]
                din, sums_next = oh.conv_tap(dc_i, wpt[i], T, d, w, w, dgrad=True, out=dcat[:, i * w:(i + 1) * w],
                                             bn=(S["r"][i - 1], dcat[:, (i - 1) * w:i * w], st_p[0], st_p[1]))
            else:
                din, sums_next = oh.conv_tap(dc_i, wpt[i], T, d, w, w, dgrad=True, out=dcat[:, i * w:(i + 1) * w]), None
            din_next = din if i > 0 else None
        # the K = 3 weight gradients of all branches in one launch: bf16 MFMA on the resident operands (exact
        # products, fp32 sums = the fp32 contraction of the widened operands, without widened copies)
        on_side(lambda: oh.conv_tap_wgrad([S["t"][i] for i in range(nums)], dcs, T, d,
                                          [gv("convs.%d.weight" % i) for i in range(nums)]), *dcs)
        st1 = S["st1"]
        dc1 = oh.bn_bwd(S["r1"], dcat, T, st1[0], st1[1], det(blk.bn1.weight), gv("bn1.weight"), gv("bn1.bias"), dx=dcat,
                        dbias=gv("conv1.bias"))
        on_side(lambda: oh.conv_wgrad(S["inp"], dc1, T, gv("conv1.weight")), dc1)
        # + dout: the residual branch (ecapa_tdnn.py:93), + add2: the concat gradient's slice of the previous block
        return oh.conv_pointwise(dc1, det(blk.conv1.weight), T, dgrad=True, acc=dout, acc2=add2)

    def _backward_h(self, S, dfeat, dout):
        arena = self.arena()
        G = arena.grad_views()
        accumulating = any(p.grad is not None and p.grad.data_ptr() == G[n].data_ptr() for n, p, _, _ in arena.entries)
        old = arena.grad.clone() if accumulating else None
        det = lambda p: p.detach()
        B, _, T = S["x"].shape
        C = self.C
        tail = ("fc7.weight", "fc7.bias", "bn7.weight", "bn7.bias")
        have_tail = dout is not None
        main = torch.cuda.current_stream()
        use_side = self.overlap_wgrad and not accumulating
        if use_side and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=main.device)
        side = self._side_stream if use_side else main
        batched = use_side and self.wgrad_batched
        keep, pending = [], []

        def on_side(fn, *reads):
            if not use_side:
                fn()
                return
            keep.extend(reads)
            if batched:
                pending.append(fn)
                return
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                fn()

        def flush_side():
            """batched mode: everything queued so far goes to the side stream behind ONE event of the main stream"""
            if not pending:
                return
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                for fn in pending:
                    fn()
            del pending[:]

        if dfeat is None:
            dfeat = torch.zeros_like(S["feat"])
        dfeat = dfeat.contiguous()
        if have_tail:  # CE branch through fc7/bn7 (dead under ang_iso, main_train.py:355 -> 376)
            do7 = dout.contiguous()
            if self.out_bn:
                st7 = S["st7"]
                do7, _, _ = ops.bn_bwd(S["o7"].view(B, -1, 1), do7.view(B, -1, 1), st7[0], st7[1], det(self.bn7.weight),
                                       det(self.bn7.bias), dgamma=G["bn7.weight"], dbeta=G["bn7.bias"])
                do7 = do7.view(B, -1)
            dx7, _, _ = ops.linear_bwd(S["feat"], det(self.fc7.weight), do7, True, dw=G["fc7.weight"], db=G["fc7.bias"])
            dfeat = ops.add_(dx7, dfeat)
        dp5, _, _ = ops.linear_bwd(S["p5"], det(self.fc6.weight), dfeat, True, dw=G["fc6.weight"], db=G["fc6.bias"])
        st5 = S["st5"]
        dpooled, _, _ = ops.bn_bwd(S["pooled"].view(B, -1, 1), dp5.view(B, -1, 1), st5[0], st5[1], det(self.bn5.weight),
                                   det(self.bn5.bias), dgamma=G["bn5.weight"], dbeta=G["bn5.bias"])
        x4, wts = S["x4"], S["wts"]
        dev = x4.device
        dx4 = torch.empty_like(x4)
        rows3 = torch.empty((B, x4.shape[1]), device=dev, dtype=torch.float32)
        oh.asp_bwd(x4, wts, T, S["pooled"], dpooled.view(B, -1).contiguous(), dx4, rowsum=rows3)  # wts -> d(logits)
        a0, a3 = self.attention[0], self.attention[3]
        ops.sum_rows(rows3, out=G["attention.3.bias"])  # analytically zero (softmax over T): rounding noise
        on_side(lambda: oh.conv_wgrad(S["a1n"], wts, T, G["attention.3.weight"]), wts)
        da1n = oh.conv_pointwise(wts, det(a3.weight), T, dgrad=True)
        stA = S["stA"]
        da1 = oh.bn_bwd(S["a1"], da1n, T, stA[0], stA[1], det(self.attention[2].weight), G["attention.2.weight"],
                        G["attention.2.bias"], dx=da1n, dbias=G["attention.0.bias"])
        gw0 = G["attention.0.weight"].view(128, -1)  # (128, 4608)

        def att0_wgrad():
            dwx = oh.conv_wgrad(x4, da1, T, torch.empty((128, 1536, 1), device=dev, dtype=torch.float32))
            ops.add_strided(gw0[:, :1536].unsqueeze(1), dwx.view(128, 1, 1536))

        on_side(att0_wgrad, da1)
        oh.conv_pointwise(da1, S["w_x"], T, dgrad=True, acc=dx4, out=dx4)
        dctxb = oh.row_stats(da1, T, want_std=False)[0] * float(T)  # (B,128): sum over time of d(a1)
        dctx, dwc, _ = ops.linear_bwd(S["ctx"], S["w_c"], dctxb.contiguous(), True, need_db=False)
        ops.add_strided(gw0[:, 1536:].unsqueeze(1), dwc.view(128, 1, 3072))
        dmean = dctx[:, :1536].contiguous()
        dstd = dctx[:, 1536:].contiguous()
        rows = torch.empty((B, x4.shape[1]), device=dev, dtype=torch.float32)
        oh.row_stats_bwd(x4, T, S["mean"], S["std"], dmean, dstd, dx4, accumulate=True, relu_mask=True, rowsum=rows)
        ops.sum_rows(rows, out=G["layer4.bias"])
        on_side(lambda: oh.conv_wgrad(S["cat123"], dx4, T, G["layer4.weight"]), dx4)
        bucketer = None if accumulating else getattr(self, "_bucketer", None)
        offsets = {n: o for n, _, o, _ in arena.entries}

        def grads_final_from(first_param):
            cut = getattr(self, "_segment_cut", None)
            if cut is not None:  # train.Trainer's segmented hipGraph capture: a segment may end here
                cut(offsets[first_param])
            if bucketer is not None:
                flush_side()
                evs = [torch.cuda.Event()]
                evs[0].record(main)
                if use_side:
                    evs.append(torch.cuda.Event())
                    evs[1].record(side)
                bucketer.ready(offsets[first_param], evs)

        if bucketer is not None:
            bucketer.reset(arena.grad, arena.head_total)
        grads_final_from("layer4.weight")
        dcat123 = oh.conv_pointwise(dx4, det(self.layer4.weight), T, dgrad=True)
        dnext = None
        for k in (2, 1, 0):
            # d(block k output) = its slice of the concat gradient + d(block k + 1 input): block k + 1's last dgrad
            # already added this block's slice (add2); block 3 reads its slice in place
            dblk = dcat123[:, k * C:(k + 1) * C] if dnext is None else dnext
            add2 = dcat123[:, (k - 1) * C:k * C] if k > 0 else None
            dnext = self._block_bwd_h(S["blocks"][k], dblk, T, G, "layer%d." % (k + 1), add2, on_side, flush_side)
            grads_final_from("layer%d.conv1.weight" % (k + 1))
        st0 = S["st0"]
        dc0 = oh.bn_bwd(S["r0"], dnext, T, st0[0], st0[1], det(self.bn1.weight), G["bn1.weight"], G["bn1.bias"], dx=dnext,
                        dbias=G["conv1.bias"])

        def conv1_wgrad():
            R0, nk = self._conv1_rows(), self.conv1.in_channels * self.conv1.kernel_size[0]
            dwm = oh.conv_wgrad(S["xcol"], dc0, T, torch.empty((C, R0, 1), device=dev, dtype=torch.float32))
            ops.add_strided(G["conv1.weight"].view(C, 1, nk), dwm.view(C, 1, R0)[:, :, :nk])  # drop the zero columns

        on_side(conv1_wgrad, dc0)
        flush_side()
        if use_side:
            main.wait_stream(side)
        del keep[:]
        arena.tail_has_grad = have_tail
        if accumulating:
            if not have_tail:
                arena.grad[arena.head_total:].zero_()
            ops.add_(arena.grad, old)
            arena.tail_has_grad = True
            return [None if (p.grad is not None and p.grad.data_ptr() == G[n].data_ptr())
                    else (G[n] if (have_tail or n not in tail) else None) for n, p, _, _ in arena.entries]
        return [G[n] if (have_tail or n not in tail) else None for n, _, _, _ in arena.entries]
