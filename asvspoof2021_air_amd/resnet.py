"""Drop-in for the reference's ``resnet.ResNet`` / ``model.ResNet`` (resnet.py:122-191)
with ``SelfAttention`` (resnet.py:11-46) and ``PreActBlock`` (resnet.py:49-69).

Same constructor, ``forward(x:(B,1,60,T)) -> (feat:(B,enc_dim), mu:(B,nclasses))``,
``state_dict`` keys (117 for ResNet-18) and construction order (so a seeded
construction consumes the torch RNG like the reference, including the discarded
``downsample`` modules of resnet.py:162-166).

The whole forward and backward run in hand-written gfx950 kernels
(csrc/conv2d.hip, norm_act.hip, pool_head.hip) reached through the C-ABI; this
file only sequences them.  One ``torch.autograd.Function`` spans the model, so
``loss.backward()`` works as in main_train.py:406 while the per-layer
bookkeeping stays out of the autograd engine.  Fusions used:
  * BatchNorm-apply + ReLU of every PreActBlock conv input is folded into the
    conv's LDS staging (forward AND wgrad), so activated tensors are never
    written to HBM;
  * the residual add (resnet.py:68) is the conv epilogue;
  * the identity-shortcut gradient join is the BatchNorm-backward epilogue;
  * the 1x1-shortcut gradient join is the dgrad epilogue.
"""
import os

import torch
import torch.nn as nn
import torch.nn.init as init

from . import _hip, ops
from .arena import ParamArena


class SelfAttention(nn.Module):
    """Parameter holder + standalone forward for resnet.py:11-46."""

    def __init__(self, hidden_size, mean_only=False):
        super().__init__()
        self.hidden_size = hidden_size
        self.att_weights = nn.Parameter(torch.Tensor(1, hidden_size), requires_grad=True)
        self.mean_only = mean_only
        init.kaiming_uniform_(self.att_weights)

    def forward(self, inputs, noise=None):
        """inputs: (B, T, H) like the reference.  Inference-only convenience path
        (the training path goes through ResNet.forward)."""
        x = inputs.permute(0, 2, 1).contiguous()
        if self.mean_only:  # resnet.py:43-44: the attention-weighted sum alone (no noise is drawn, no std)
            out, _ = ops.selfatt_pool_fwd(x, self.att_weights.detach(), None)
            return out[:, :self.hidden_size].contiguous()
        out, _ = ops.selfatt_pool_fwd(x, self.att_weights.detach(), noise)
        return out


class PreActBlock(nn.Module):
    """Parameter holder for the pre-activation BasicBlock (resnet.py:49-69)."""
    expansion = 1

    def __init__(self, in_planes, planes, stride, *args, **kwargs):
        super().__init__()
        self.stride = stride
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False))

    def forward(self, x):
        """Stand-alone forward of resnet.py:63-69 composed from the same kernels ``ResNet.forward`` sequences
        (BatchNorm statistics -> apply + ReLU -> 1x1 shortcut on the activated tensor -> conv1 -> BatchNorm + ReLU
        -> conv2 + shortcut in the epilogue).  Forward only: training through a lone block is not on the hot path
        (nobody in the reference calls a block directly), so a graph-recording call raises."""
        if not x.is_cuda:
            raise _hip.AirError("PreActBlock HIP path needs a GPU tensor; there is no CPU fallback")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("PreActBlock.forward is forward-only (use torch.no_grad()); gradients flow "
                                      "through ResNet.forward")
        x = x.float().contiguous()
        w = lambda conv: conv.weight.detach()
        st1 = _bn_train_coeffs(x, self.bn1, self.training)
        a1 = ops.bn_apply(x, st1[2], st1[3], relu=True)
        sc = ops.conv2d_fwd(a1, w(self.shortcut[0]), self.stride, 0) if hasattr(self, "shortcut") else x
        h = ops.conv2d_fwd(a1, w(self.conv1), self.stride, 1)
        st2 = _bn_train_coeffs(h, self.bn2, self.training)
        out = ops.conv2d_fwd(ops.bn_apply(h, st2[2], st2[3], relu=True), w(self.conv2), 1, 1, residual=sc)
        ops.bn_flush()
        return out


RESNET_CONFIGS = {"18": [[2, 2, 2, 2], PreActBlock],
                  "28": [[3, 4, 6, 3], PreActBlock],
                  "34": [[3, 4, 6, 3], PreActBlock]}


def _bn_train_coeffs(x, bn, training, stats_in=None):
    """(mean, invstd, scale, shift) for BatchNorm ``bn`` on ``x``.  stats_in: statistics records of ``x`` from the
    epilogue of the convolution that produced it (ops.conv2d_fwd(..., stats=True)), or None."""
    if training:
        mean, invstd, scale, shift = ops.bn_stats(x, bn.weight.detach(), bn.bias.detach(),
                                                  bn.running_mean, bn.running_var, bn.eps,
                                                  bn.momentum, stats_in=stats_in)
        ops.bn_tick(bn.num_batches_tracked)
        return mean, invstd, scale, shift
    scale, shift = ops.bn_eval_coeffs(bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                      bn.running_var, bn.eps)
    return None, None, scale, shift


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, noise, *params):
        ctx.set_materialize_grads(False)
        feat, mu, saved = model._forward_impl(x, noise, save=True)
        ctx.model = model
        ctx.saved = saved
        if getattr(model, "keep_saved_for_test", False):  # parity tests read the ReLU decisions of the step
            model._last_saved_for_test = saved
        return feat, mu

    @staticmethod
    def backward(ctx, dfeat, dmu):
        model, saved = ctx.model, ctx.saved
        ctx.saved = None
        grads = model._backward_impl(saved, dfeat, dmu)
        return (None, None, None) + tuple(grads)


class ResNet(nn.Module):
    MAX_POOL_FRAMES = 148  # csrc/pool_head.hip: (256 (T' + 1) + 2 T') floats <= 150 KB of LDS: the LDS-resident kernel
    MAX_POOL_FRAMES_GLOBAL = 12000  # its in-place variant: (3 T' + 3 x 256) floats of per-frame vectors in LDS

    def __init__(self, num_nodes, enc_dim, resnet_type="18", nclasses=2):
        self.in_planes = 16
        super().__init__()
        layers, block = RESNET_CONFIGS[resnet_type]
        self._norm_layer = nn.BatchNorm2d
        self.conv1 = nn.Conv2d(1, 16, kernel_size=(9, 3), stride=(3, 1), padding=(1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        self.activation = nn.ReLU()
        self.layer1 = self._make_layer(block, 64, layers[0], stride=1)
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.conv5 = nn.Conv2d(512 * block.expansion, 256, kernel_size=(num_nodes, 3), stride=(1, 1),
                               padding=(0, 1), bias=False)
        self.bn5 = nn.BatchNorm2d(256)
        self.fc = nn.Linear(256 * 2, enc_dim)
        self.fc_mu = nn.Linear(enc_dim, nclasses) if nclasses >= 2 else nn.Linear(enc_dim, 1)
        self.initialize_params()
        self.attention = SelfAttention(256)
        # attention noise (resnet.py:38): 'device' = on-GPU Philox draw, 'none', or a tensor
        # installed with set_attention_noise() (parity tests replay the reference's draw)
        self.noise_mode = "device"
        self.noise_scale = 1e-5
        self._noise_tensor = None
        self._noise_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self._noise_offset = 0
        self._noise_ctr = None  # the offset of the Philox stream as a device-side counter (ops.randn_ctr)
        self._arena = None
        # True: BatchNorm-apply + ReLU folded into every conv's operand read (no activated
        # tensor in HBM).  False: one HBM-bound pass writes the activated tensor and the convs
        # run their plain (faster) MFMA loop.  Measured on MI355X: see DESIGN.md §4.
        self.fuse_bn_into_conv = os.environ.get("AIR_FUSE_BN", "0") == "1"
        # BatchNorm batch statistics out of the producing convolution's epilogue (AIR_BN_STATS=0: every BatchNorm
        # reduces its input itself, rounds 1-3)
        self.fuse_bn_stats = os.environ.get("AIR_BN_STATS", "1") == "1"
        # weight gradients on a side HIP stream, overlapping the HBM-bound BN-backward passes
        self.overlap_wgrad = os.environ.get("AIR_OVERLAP_WGRAD", "1") == "1"
        self._side_stream = None
        # Winograd weight transforms (a ~9 us kernel in front of each of the 16 + 16 forward / dgrad launches of the
        # 3x3 stride-1 layers) depend on the weights only: from the second training step on they run on the side
        # stream at the start of the step, under the front-end and the first layers (ops.conv2d_prepack)
        self.prepack_weights = os.environ.get("AIR_PREPACK_WEIGHTS", "1") == "1"
        self._geo = {}    # layer key -> (input shape, stride, padding) seen by the last training forward
        self._packs = {}  # (layer key, pass) -> persistent buffer
        self._pack_ev = [None, None]
        self._bucketer = None  # dist.GradBucketer when the all-reduce is overlapped with backward

    def enable_ddp_overlap(self, bucket_bytes=None):
        """Launch the gradient all-reduce from inside backward (one process per GPU, world size > 1)."""
        from .dist import GradBucketer
        self._bucketer = GradBucketer(bucket_bytes)
        return self

    def __getstate__(self):
        """Whole-module pickles (main_train.py:675-704 -> generate_score.py:46-48): the flat arenas, the
        side stream and an installed noise tensor are runtime state and are rebuilt on first use."""
        st = dict(self.__dict__)
        st["_arena"] = None
        st["_side_stream"] = None
        st["_geo"], st["_packs"], st["_pack_ev"] = {}, {}, [None, None]
        st["_bucketer"] = None
        st["_segment_cut"] = None
        st["_noise_tensor"] = None
        if st.get("_noise_ctr") is not None:  # the device-side counter travels as its value
            st["_noise_offset"] = int(st["_noise_ctr"].item())
        st["_noise_ctr"] = None
        st["_noise_ctrs"] = None
        st.pop("_last_saved_for_test", None)
        st.pop("keep_saved_for_test", None)
        if st.get("noise_mode") == "tensor":
            st["noise_mode"] = "device"
        return st

    def initialize_params(self):
        """resnet.py:149-157."""
        for layer in self.modules():
            if isinstance(layer, nn.Conv2d):
                init.kaiming_normal_(layer.weight, a=0, mode="fan_out")
            elif isinstance(layer, nn.Linear):
                init.kaiming_uniform_(layer.weight)
            elif isinstance(layer, (nn.BatchNorm2d, nn.BatchNorm1d)):
                layer.weight.data.fill_(1)
                layer.bias.data.zero_()

    def _make_layer(self, block, planes, num_blocks, stride=1):
        # the reference builds a conv1x1+BN ``downsample`` here and drops it (resnet.py:162-166):
        # build and discard it too so a seeded construction draws the same random numbers
        if stride != 1 or self.in_planes != planes * block.expansion:
            nn.Sequential(nn.Conv2d(self.in_planes, planes * block.expansion, 1, stride, bias=False),
                          nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.in_planes, planes, stride)]
        self.in_planes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(self.in_planes, planes, 1))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ plumbing
    def blocks(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                yield blk

    def arena(self):
        """Flat parameter/gradient arenas (built lazily, rebuilt after .to(device))."""
        dev = self.conv1.weight.device
        if self._arena is None:
            self._arena = ParamArena(list(self.named_parameters()),
                                     tail_names=("fc_mu.weight", "fc_mu.bias"))
        if not self._arena.bound() or self._arena.device != dev:
            self._arena.bind(dev)
        return self._arena

    def set_attention_noise(self, noise):
        """Install the (B, T', 256) noise tensor to use (already scaled), or None."""
        self._noise_tensor = noise
        self.noise_mode = "tensor" if noise is not None else "none"

    def _draw_noise(self, B, T, device):
        if self.noise_mode == "none":
            return None
        if self.noise_mode == "tensor":
            n = self._noise_tensor
            if tuple(n.shape) != (B, T, 256):
                raise _hip.AirError("attention noise must be (B, T', 256), got %s" % (tuple(n.shape),))
            return n.to(device).contiguous()
        # (seed, offset) of the Philox stream: the offset lives on the device and the draw advances it there, so
        # that a step captured in a hipGraph draws fresh noise on every replay (resnet.py:38 draws per call) and
        # eager launches and replays walk one sequence
        # ONE counter per device, never replaced once made: a captured hipGraph (train.Trainer, GraphedScorer) holds
        # its address.  Moving to another device first folds the live count back into the host field, so that the
        # Philox sequence goes on instead of restarting (it restarted from the stale host value).
        ctrs = getattr(self, "_noise_ctrs", None)
        if ctrs is None:
            ctrs = self._noise_ctrs = {}
        ctr = ctrs.get(device)
        if ctr is None:
            live = getattr(self, "_noise_ctr", None)
            if live is not None:
                self._noise_offset = int(live.item())
            ctr = ctrs[device] = torch.tensor([self._noise_offset], dtype=torch.int64, device=device)
        elif getattr(self, "_noise_ctr", None) is not ctr:
            # back on a device used before: carry the count of the counter used in between over (device-side copy)
            live = getattr(self, "_noise_ctr", None)
            if live is not None:
                ctr.copy_(live)
        self._noise_ctr = ctr
        return ops.randn_ctr((B, T, 256), device, self._noise_seed, ctr, self.noise_scale)

    def forward(self, x):
        if not x.is_cuda:
            raise _hip.AirError("ResNet HIP path needs a GPU tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("ResNet expects (B, 1, F, T), got %s" % (tuple(x.shape),))
        if x.shape[0] == 1 and self.training:
            pass  # the reference special-cases B==1 in SelfAttention (resnet.py:28-30); same maths here
        ta = x.shape[3]
        for _ in range(3):  # the three stride-2 stages (resnet.py:136-138)
            ta = (ta - 1) // 2 + 1
        if ta > self.MAX_POOL_FRAMES_GLOBAL:
            # checked up front (not in the middle of a scoring run).  Up to MAX_POOL_FRAMES pooled frames the
            # attention pooling kernel keeps an utterance's (256, T') map in LDS; longer ones (the reference pools
            # any length, resnet.py:23-46) take its in-place variant, whose per-frame vectors still live in LDS
            raise ValueError("ResNet HIP path: %d input frames give %d pooled frames; the SelfAttention pooling "
                             "kernels hold at most %d" % (x.shape[3], ta, self.MAX_POOL_FRAMES_GLOBAL))
        x = x.float().contiguous()  # main_train.py:338 hands over a transposed view
        arena = self.arena()
        # eval-mode forward never records a graph (backward through running-stat BN is not
        # on the hot path; generate_score.py only scores)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for _, p, _, _ in arena.entries):
            params = [p for _, p, _, _ in arena.entries]
            return _ResNetFn.apply(self, x, None, *params)
        feat, mu, _ = self._forward_impl(x, None, save=False)
        return feat, mu

    def forward_saved(self, x):
        """The train-mode forward WITHOUT autograd: (feat, saved).  With ``backward_saved`` this is what
        ``_ResNetFn`` does, callable from one Python thread - train.Trainer captures the step as several hipGraphs cut
        between backward's bucket boundaries (autograd would run backward on its own worker thread)."""
        x = x.float().contiguous()
        self.arena()
        feat, _, saved = self._forward_impl(x, None, save=True)
        return feat, saved

    def backward_saved(self, saved, dfeat):
        """Gradients of every arena entry (views of the gradient arena, None where there is none), in arena order."""
        return self._backward_impl(saved, dfeat, None)

    def _launch_prepack(self, fuse):
        """Enqueue the weight transforms of every conv behind conv1 - Winograd for the 3x3 stride-1 layers, the direct
        kernels' slabs for the rest (forward unless the BatchNorm is fused into the conv's operand read, dgrad
        always) - on the side stream, for the layer geometries the previous training forward saw.  Returns
        {(layer key, pass): buffer}; events in self._pack_ev."""
        self._geo_live = {}
        if not self._geo:
            return {}
        main = torch.cuda.current_stream()
        # one chain (overlap_wgrad off: the hipGraph capture of train.Trainer): the transforms run on the main stream
        # in front of the first layer that needs them, still as one launch per 32
        one_chain = not self.overlap_wgrad
        if not one_chain and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=main.device)
        side = main if one_chain else self._side_stream
        if not one_chain:
            start = torch.cuda.Event()
            start.record(main)  # the optimiser step that produced these weights is in front of it
            side.wait_event(start)
        live = {}
        blocks = list(self.blocks())
        with torch.cuda.stream(side):
            for which in (0, 1):
                if which == 0 and fuse:
                    continue
                with ops.prepack_batch():  # one launch per 32 transforms instead of one per layer
                    for (bi, ci), (shape, s, pad) in self._geo.items():
                        # (bi, 1 | 2): a block's 3x3 convs; (bi, 0): its 1x1 shortcut; (-1, 5): conv5
                        conv = self.conv5 if bi < 0 else (blocks[bi].shortcut[0] if ci == 0 else
                                                          blocks[bi].conv1 if ci == 1 else blocks[bi].conv2)
                        if which == 0 and bi >= 0 and s == 2 and ci in (0, 1) and hasattr(blocks[bi], "shortcut") and \
                                ops.conv2d_fwd_s2_pair_ok(blocks[bi].conv1.weight.shape, shape):
                            # a stride-2 block's conv1 and shortcut share ONE forward launch (round 5): their weights
                            # travel together under (bi, 1); the shortcut has no buffer of its own then
                            if ci == 0:
                                continue
                            buf = ops.conv2d_fwd_s2_pair_prepack(conv.weight.detach(), blocks[bi].shortcut[0].weight.detach(),
                                                                 shape, out=self._packs.get(((bi, ci), "fpair")))
                            if buf is not None:
                                self._packs[((bi, ci), "fpair")] = buf
                                live[((bi, ci), which)] = buf
                                self._geo_live[(bi, ci)] = shape
                                continue
                        if which == 1 and bi >= 0 and s == 2 and ci in (0, 1) and hasattr(blocks[bi], "shortcut"):
                            # a stride-2 block's conv1 and shortcut share ONE data-gradient launch (round 4): their
                            # weights travel together under (bi, 1); the shortcut has no buffer of its own then
                            if ci == 0 and ops.conv2d_dgrad_s2_pair_ok(blocks[bi].conv1.weight.shape, shape):
                                continue
                            if ci == 1:
                                buf = ops.conv2d_dgrad_s2_pair_prepack(conv.weight.detach(), blocks[bi].shortcut[0].weight.detach(),
                                                                       shape, out=self._packs.get(((bi, ci), "pair")))
                                if buf is not None:
                                    self._packs[((bi, ci), "pair")] = buf
                                    live[((bi, ci), which)] = buf
                                    self._geo_live[(bi, ci)] = shape
                                    continue
                        buf = ops.conv2d_prepack(conv.weight.detach(), shape, s, pad, which, out=self._packs.get(((bi, ci), which)))
                        if buf is not None:
                            self._packs[((bi, ci), which)] = buf
                            live[((bi, ci), which)] = buf
                            self._geo_live[(bi, ci)] = shape
                if one_chain:
                    self._pack_ev[which] = None
                else:
                    self._pack_ev[which] = torch.cuda.Event()
                    self._pack_ev[which].record(side)
        return live

    # ------------------------------------------------------------------ forward
    def _forward_impl(self, x, noise, save):
        training = self.training
        S = {} if save else None
        w = lambda conv: conv.weight.detach()
        c1 = ops.conv2d_fwd(x, w(self.conv1), (3, 1), (1, 1))  # resnet.py:176
        st1 = _bn_train_coeffs(c1, self.bn1, training)
        cur = ops.bn_apply(c1, st1[2], st1[3], relu=True)  # resnet.py:177
        if save:
            S["x"], S["c1"], S["st1"] = x, c1, st1
            S["blocks"] = []
        fuse = self.fuse_bn_into_conv
        prepack = self.prepack_weights and save and training
        live = self._launch_prepack(fuse) if prepack else {}
        waited = [False]

        def packed(key, which, shape):
            """Buffer of transformed weights for this call, or None (first step, other shapes, non-Winograd layer)."""
            if prepack:
                self._geo[key] = (tuple(shape), self._geo_sp[key][0], self._geo_sp[key][1])
            buf = live.get((key, which))
            if buf is None or self._geo_live.get(key) != tuple(shape):
                return None
            if which == 0 and not waited[0]:
                if self._pack_ev[0] is not None:
                    torch.cuda.current_stream().wait_event(self._pack_ev[0])
                waited[0] = True
            return buf

        self._geo_sp = {}
        # BatchNorm statistics from the producing convolution's epilogue (round 4): conv1 -> bn2 and block output ->
        # the next block's bn1, wherever the convolution is a Winograd F(3x4, 3x3) launch (ops returns None otherwise)
        want_stats = training and getattr(self, "fuse_bn_stats", True)
        cur_rec = None
        blocks = list(self.blocks())

        def conv_st(want, *a, **kw):  # (y, records or None)
            return ops.conv2d_fwd(*a, stats=True, **kw) if want else (ops.conv2d_fwd(*a, **kw), None)

        for bi, blk in enumerate(blocks):
            s = blk.stride
            self._geo_sp[(bi, 1)], self._geo_sp[(bi, 2)], self._geo_sp[(bi, 0)] = (s, 1), (1, 1), (s, 0)
            stA = _bn_train_coeffs(cur, blk.bn1, training, cur_rec)
            if fuse:  # BN-apply + ReLU folded into the conv's operand read (no activated tensor)
                actA, pA = cur, dict(in_scale=stA[2], in_shift=stA[3], relu=True)
            else:     # activated tensor written once (HBM-bound pass), convs run their plain loop
                actA, pA = ops.bn_apply(cur, stA[2], stA[3], relu=True), {}
            fpair = hasattr(blk, "shortcut") and s == 2 and not fuse and ops.conv2d_fwd_s2_pair_ok(blk.conv1.weight.shape, actA.shape)
            if fpair:  # conv1 and the 1x1 shortcut of a stride-2 block in one launch over actA (round 5)
                if prepack:
                    self._geo[(bi, 0)] = (tuple(actA.shape), s, 0)
                h, sc = ops.conv2d_fwd_s2_pair(actA, w(blk.conv1), w(blk.shortcut[0]), packed=packed((bi, 1), 0, actA.shape))
                h_rec = None
            else:
                if hasattr(blk, "shortcut"):
                    sc = ops.conv2d_fwd(actA, w(blk.shortcut[0]), s, 0, w_packed=packed((bi, 0), 0, actA.shape), **pA)
                else:
                    sc = cur
                h, h_rec = conv_st(want_stats, actA, w(blk.conv1), s, 1, w_packed=packed((bi, 1), 0, actA.shape), **pA)
            stB = _bn_train_coeffs(h, blk.bn2, training, h_rec)
            if fuse:
                actB, pB = h, dict(in_scale=stB[2], in_shift=stB[3], relu=True)
            else:
                actB, pB = ops.bn_apply(h, stB[2], stB[3], relu=True), {}
            # (the last block's output feeds conv5, not a BatchNorm: resnet.py:182)
            out, cur_rec = conv_st(want_stats and bi + 1 < len(blocks), actB, w(blk.conv2), 1, 1, residual=sc,
                                   w_packed=packed((bi, 2), 0, actB.shape), **pB)
            if save:
                S["blocks"].append((blk, cur, stA, h, stB, actA, actB))
            cur = out
        self._geo_sp[(-1, 5)] = (1, (0, 1))
        c5 = ops.conv2d_fwd(cur, w(self.conv5), 1, (0, 1), w_packed=packed((-1, 5), 0, cur.shape))  # resnet.py:182
        st5 = _bn_train_coeffs(c5, self.bn5, training)
        a5 = ops.bn_apply(c5, st5[2], st5[3], relu=True)  # resnet.py:183
        B, C5, H5, T5 = a5.shape
        if H5 != 1:
            raise _hip.AirError("conv5 must reduce the frequency axis to 1 (got %d): input height "
                                "does not match num_nodes" % H5)
        a5v = a5.view(B, C5, T5)
        nz = noise if noise is not None else self._draw_noise(B, T5, x.device)
        pooled, alpha = ops.selfatt_pool_fwd(a5v, self.attention.att_weights.detach(), nz)
        feat = ops.linear_fwd(pooled, self.fc.weight.detach(), self.fc.bias.detach())
        mu = ops.linear_fwd(feat, self.fc_mu.weight.detach(), self.fc_mu.bias.detach())
        if save:
            if not training:
                raise NotImplementedError("backward through eval-mode BatchNorm is not on the hot path")
            S.update(l4=cur, c5=c5, st5=st5, a5v=a5v, noise=nz, pooled=pooled, alpha=alpha, feat=feat,
                     # (data-gradient buffers only for the geometry THIS forward saw: a direct kernel's slab layout
                     # depends on the batch size through its tile heuristic - the short last batch of an epoch)
                     packs={k: v for k, v in live.items() if k[1] == 1 and self._geo_live.get(k[0]) is not None and
                            self._geo_live.get(k[0]) == self._geo.get(k[0], (None,))[0]},
                     pack_ev=self._pack_ev[1] if live else None)
        ops.bn_flush()
        return feat, mu, S

    # ----------------------------------------------------------------- backward
    def _backward_impl(self, S, dfeat, dmu):
        arena = self.arena()
        G = arena.grad_views()
        have = set()
        # gradient accumulation (backward twice without zero_grad): p.grad already IS the arena
        # view, so keep the old sums aside and fold them back in at the end
        accumulating = any(p.grad is not None and p.grad.data_ptr() == G[n].data_ptr()
                           for n, p, _, _ in arena.entries)
        old = arena.grad.clone() if accumulating else None

        def gv(mod_name):
            have.add(mod_name)
            return G[mod_name]

        names = {id(p): n for n, p in self.named_parameters()}
        nm = lambda p: names[id(p)]
        w = lambda conv: conv.weight.detach()

        if dfeat is None:
            dfeat = torch.zeros_like(S["feat"])
        dfeat = dfeat.contiguous()
        if dmu is not None:  # CE / base-loss branch (main_train.py:355); dead under ang_iso
            dmu = dmu.contiguous()
            dx_mu, _, _ = ops.linear_bwd(S["feat"], self.fc_mu.weight.detach(), dmu, True,
                                         dw=gv("fc_mu.weight"), db=gv("fc_mu.bias"))
            dfeat = ops.add_(dx_mu, dfeat)
        dpooled, _, _ = ops.linear_bwd(S["pooled"], self.fc.weight.detach(), dfeat, True,
                                       dw=gv("fc.weight"), db=gv("fc.bias"))
        da5, datt = ops.selfatt_pool_bwd(S["a5v"], self.attention.att_weights.detach(), S["noise"],
                                         S["alpha"], S["pooled"], dpooled)
        ops.sum_rows(datt, out=gv("attention.att_weights").view(-1))
        # Weight gradients feed nothing until the optimiser, so they run on a SIDE stream: the
        # MFMA-bound wgrad kernels overlap the HBM-bound BatchNorm-backward passes and the tails
        # of the dgrad chain on the main stream.  Ordering: a wgrad starts after the event that
        # marks its dy ready; an in-place update of a tensor a wgrad still reads waits for that
        # wgrad's event; the main stream joins the side stream before the gradients are used.
        main = torch.cuda.current_stream()
        use_side = self.overlap_wgrad
        if use_side and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=main.device)
        side = self._side_stream if use_side else main
        keep = []  # tensors the side stream reads: keep them alive until the join

        def on_side(fn, *reads):
            """Run fn() on the side stream once everything enqueued on main so far is done.
            Returns an event marking its completion."""
            if not use_side:
                fn()
                return None
            keep.extend(reads)
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                fn()
                done = torch.cuda.Event()
                done.record(side)
            return done

        def wait_for(ev):
            if ev is not None:
                main.wait_event(ev)

        offsets = {n: o for n, _, o, _ in arena.entries}
        bucketer = getattr(self, "_bucketer", None)
        if accumulating:
            bucketer = None
        if bucketer is not None:
            bucketer.reset(arena.grad, arena.head_total)

        def grads_final_from(first_param):
            """Everything that writes arena.grad[offset(first_param):] has been enqueued."""
            cut = getattr(self, "_segment_cut", None)
            if cut is not None:  # train.Trainer's segmented hipGraph capture: a segment may end here
                cut(offsets[first_param])
            if bucketer is None:
                return
            evs = [torch.cuda.Event()]
            evs[0].record(main)
            if use_side:
                evs.append(torch.cuda.Event())
                evs[1].record(side)
            bucketer.ready(offsets[first_param], evs)

        c5, st5 = S["c5"], S["st5"]
        dc5, _, _ = ops.bn_bwd(c5, da5.view_as(c5), st5[0], st5[1], self.bn5.weight.detach(),
                               self.bn5.bias.detach(), relu=True,
                               dgamma=gv("bn5.weight"), dbeta=gv("bn5.bias"))
        l4 = S["l4"]
        g5 = gv("conv5.weight")
        on_side(lambda: ops.conv2d_wgrad(l4, dc5, self.conv5.weight.shape, 1, (0, 1), out=g5), dc5)
        grads_final_from("conv5.weight")
        fuse = self.fuse_bn_into_conv
        packs = S.get("packs") or {}
        if packs and S.get("pack_ev") is not None:
            main.wait_event(S["pack_ev"])  # the dgrad weight transforms enqueued on the side stream during forward
        dcur = ops.conv2d_dgrad(dc5, w(self.conv5), l4.shape, 1, (0, 1), w_packed=packs.get(((-1, 5), 1)))
        nblk = len(S["blocks"])

        def dgrad_bn(bn, *a, **kw):  # (dx, BatchNorm-backward sums or None)
            return ops.conv2d_dgrad(*a, bn=bn, **kw) if bn is not None else (ops.conv2d_dgrad(*a, **kw), None)

        for ri, (blk, xin, stA, h, stB, actA, actB) in enumerate(reversed(S["blocks"])):
            bi = nblk - 1 - ri
            s = blk.stride
            pre = nm(blk.conv1.weight)[:-len("conv1.weight")]
            pA = dict(in_scale=stA[2], in_shift=stA[3], relu=True) if fuse else {}
            pB = dict(in_scale=stB[2], in_shift=stB[3], relu=True) if fuse else {}
            has_sc = hasattr(blk, "shortcut")
            # out = conv2(actB(h)) + shortcut: both weight gradients read dcur
            g2 = gv(pre + "conv2.weight")
            gsc = gv(pre + "shortcut.0.weight") if has_sc else None

            def wg_out(dcur=dcur, actB=actB, actA=actA, blk=blk, s=s, pA=pA, pB=pB, g2=g2, gsc=gsc,
                       has_sc=has_sc):
                ops.conv2d_wgrad(actB, dcur, blk.conv2.weight.shape, 1, 1, out=g2, **pB)
                if has_sc:
                    ops.conv2d_wgrad(actA, dcur, blk.shortcut[0].weight.shape, s, 0, out=gsc, **pA)

            ev_dcur = on_side(wg_out, dcur)
            # d(relu(bn2(h))) from conv2's data gradient - whose epilogue also takes the two sums of bn2's backward
            # (round 4: ops returns None where the layer has no Winograd data gradient)
            bn_fuse = getattr(self, "fuse_bn_stats", True) and not fuse
            bnB = (h, stB[0], stB[1], blk.bn2.weight.detach(), blk.bn2.bias.detach()) if bn_fuse else None
            d_actB, smB = dgrad_bn(bnB, dcur, w(blk.conv2), h.shape, 1, 1, w_packed=packs.get(((bi, 2), 1)))
            dh, _, _ = ops.bn_bwd(h, d_actB, stB[0], stB[1], blk.bn2.weight.detach(),
                                  blk.bn2.bias.detach(), relu=True, dx=d_actB, sums_in=smB,
                                  dgamma=gv(pre + "bn2.weight"), dbeta=gv(pre + "bn2.bias"))
            g1 = gv(pre + "conv1.weight")
            on_side(lambda dh=dh, actA=actA, blk=blk, s=s, pA=pA, g1=g1:
                    ops.conv2d_wgrad(actA, dh, blk.conv1.weight.shape, s, 1, out=g1, **pA), dh)
            # (with a 1x1 shortcut d_actA gets a second term below: the sums would be of a partial gradient)
            bnA = (xin, stA[0], stA[1], blk.bn1.weight.detach(), blk.bn1.bias.detach()) if bn_fuse and not has_sc else None
            pair = has_sc and s == 2 and ops.conv2d_dgrad_s2_pair_ok(blk.conv1.weight.shape, xin.shape)
            if pair:  # conv1's and the shortcut's data gradients in one pass over (dh, dcur)
                d_actA, smA = ops.conv2d_dgrad_s2_pair(dh, w(blk.conv1), dcur, w(blk.shortcut[0]), xin.shape,
                                                       packed=packs.get(((bi, 1), 1))), None
            else:
                d_actA, smA = dgrad_bn(bnA, dh, w(blk.conv1), xin.shape, s, 1, w_packed=packs.get(((bi, 1), 1)))
            if has_sc:
                if not pair:
                    ops.conv2d_dgrad(dcur, w(blk.shortcut[0]), xin.shape, s, 0, accumulate=d_actA,
                                     out=d_actA, w_packed=packs.get(((bi, 0), 1)))
                dcur, _, _ = ops.bn_bwd(xin, d_actA, stA[0], stA[1], blk.bn1.weight.detach(),
                                        blk.bn1.bias.detach(), relu=True, dx=d_actA,
                                        dgamma=gv(pre + "bn1.weight"), dbeta=gv(pre + "bn1.bias"))
            else:
                # identity shortcut: d(block input) = bn1-backward(d_actA) + dcur, joined in place
                # (dcur is still being read by this block's conv2 wgrad on the side stream)
                wait_for(ev_dcur)
                dcur, _, _ = ops.bn_bwd(xin, d_actA, stA[0], stA[1], blk.bn1.weight.detach(),
                                        blk.bn1.bias.detach(), relu=True, dx=dcur, accumulate=True, sums_in=smA,
                                        dgamma=gv(pre + "bn1.weight"), dbeta=gv(pre + "bn1.bias"))
            grads_final_from(pre + "bn1.weight")  # the block's first parameter in arena order
        c1, st1 = S["c1"], S["st1"]
        dc1, _, _ = ops.bn_bwd(c1, dcur, st1[0], st1[1], self.bn1.weight.detach(),
                               self.bn1.bias.detach(), relu=True, dx=dcur,
                               dgamma=gv("bn1.weight"), dbeta=gv("bn1.bias"))
        gc1 = gv("conv1.weight")
        ev = on_side(lambda: ops.conv2d_wgrad(S["x"], dc1, self.conv1.weight.shape, (3, 1), (1, 1), out=gc1),
                     dc1)
        if use_side:
            main.wait_stream(side)  # join: every weight gradient is in the arena
        keep.clear()
        arena.tail_has_grad = "fc_mu.weight" in have
        if accumulating:
            ops.add_(arena.grad, old)
            return [None if (p.grad is not None and p.grad.data_ptr() == G[n].data_ptr())
                    else (G[n] if n in have else None) for n, p, _, _ in arena.entries]
        return [G[n] if n in have else None for n, _, _, _ in arena.entries]
