"""Drop-in for the adversarial channel-classifier branch of the reference (``--ADV_AUG``):
``model.GradientReversal`` / ``model.ChannelClassifier`` (model.py:976-1023), ``nn.CrossEntropyLoss``
as used at main_train.py:251, and the two-phase step of main_train.py:377-403 + :420-453.

Same constructors, ``state_dict`` keys (``classifier.0.*``, ``classifier.3.*``) and forward
semantics; every device computation is a HIP kernel behind the C-ABI (csrc/adv_head.hip plus the
linear kernels), one ``autograd.Function`` per module.  ``nn.Dropout(0.3)`` draws its mask on the
device (Philox4x32-10, seeded per module); tests pass an explicit mask to replay the reference's.
"""
import ctypes

import torch
import torch.distributed as td
import torch.nn as nn
import torch.nn.init as init

from . import _hip, ops
from . import dist as air_dist
from ._hip import ci, cf, csz, dptr, stream
from .train import Trainer, adjust_learning_rate


def _n(t):
    return csz(t.numel())


def dropout_mask(shape, p, seed, offset, device):
    keep = torch.empty(shape, device=device, dtype=torch.float32)
    _hip.check(_hip.lib().air_dropout_mask(dptr(keep), _n(keep), cf(p), ctypes.c_uint64(seed),
                                           ctypes.c_uint64(offset), stream()), "air_dropout_mask")
    return keep


def _mask_relu_fwd(x, keep):
    y = torch.empty_like(x)
    _hip.check(_hip.lib().air_mask_relu_fwd(dptr(x), dptr(keep, allow_none=True), _n(x), dptr(y), stream()),
               "air_mask_relu_fwd")
    return y


def _mask_relu_bwd(dy, y, keep, alpha=1.0):
    dx = torch.empty_like(dy)
    _hip.check(_hip.lib().air_mask_relu_bwd(dptr(dy), dptr(y), dptr(keep, allow_none=True), _n(dy), cf(alpha),
                                            dptr(dx), stream()), "air_mask_relu_bwd")
    return dx


def scale_(x, alpha):
    _hip.check(_hip.lib().air_scale(dptr(x), _n(x), cf(alpha), stream()), "air_scale")
    return x


class _GRLFn(torch.autograd.Function):
    """model.py:976-995: identity forward, dx = -lambda * dy."""

    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = float(lambda_)
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return scale_(g.contiguous().clone(), -ctx.lambda_), None


class GradientReversal(nn.Module):
    def __init__(self, lambda_=1):
        super().__init__()
        self.lambda_ = lambda_

    def forward(self, x):
        if not x.is_cuda:
            raise _hip.AirError("GradientReversal HIP path needs a GPU tensor; there is no CPU fallback")
        return _GRLFn.apply(x, self.lambda_)


class _ClassifierFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, keep, lambda_):
        x = x.contiguous()
        h = _mask_relu_fwd(ops.linear_fwd(x, w1.detach(), b1.detach()), keep)  # Linear -> Dropout -> ReLU
        o = ops.linear_fwd(h, w2.detach(), b2.detach(), relu=True)            # Linear -> ReLU
        ctx.save_for_backward(x, w1, w2, h, o)
        ctx.keep, ctx.lambda_ = keep, float(lambda_)
        return o

    @staticmethod
    def backward(ctx, do):
        x, w1, w2, h, o = ctx.saved_tensors
        do_pre = _mask_relu_bwd(do.contiguous(), o, None)
        dh, dw2, db2 = ops.linear_bwd(h, w2.detach(), do_pre)
        dh_pre = _mask_relu_bwd(dh, h, ctx.keep)
        dx, dw1, db1 = ops.linear_bwd(x, w1.detach(), dh_pre)
        scale_(dx, -ctx.lambda_)  # gradient reversal (model.py:993)
        return dx, dw1, db1, dw2, db2, None, None


class ChannelClassifier(nn.Module):
    """model.py:998-1023."""

    def __init__(self, enc_dim, nclasses, lambda_):
        super().__init__()
        self.grl = GradientReversal(lambda_)
        self.classifier = nn.Sequential(nn.Linear(enc_dim, enc_dim // 2),
                                        nn.Dropout(0.3),
                                        nn.ReLU(),
                                        nn.Linear(enc_dim // 2, nclasses),
                                        nn.ReLU())
        self._seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self._offset = 0

    def initialize_params(self):
        for layer in self.modules():
            if isinstance(layer, torch.nn.Linear):
                init.kaiming_uniform_(layer.weight)

    def forward(self, x, keep=None):
        """x (B, enc_dim) GPU features.  ``keep``: optional explicit dropout keep-mask (B, enc_dim//2),
        already scaled by 1/(1-p); default: drawn on the device in training mode, none in eval."""
        if not x.is_cuda:
            raise _hip.AirError("ChannelClassifier HIP path needs a GPU tensor; there is no CPU fallback")
        l1, l2 = self.classifier[0], self.classifier[3]
        if x.dim() != 2 or x.shape[1] != l1.in_features:
            raise ValueError("expected (B, %d) features, got %s" % (l1.in_features, tuple(x.shape)))
        p = self.classifier[1].p
        if keep is None and self.training and p > 0:
            keep = dropout_mask((x.shape[0], l1.out_features), p, self._seed, self._offset, x.device)
            self._offset += (x.shape[0] * l1.out_features + 3) // 4
        return _ClassifierFn.apply(x.float(), l1.weight, l1.bias, l2.weight, l2.bias, keep, self.grl.lambda_)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, owner):
        logits = logits.contiguous()
        B, C = logits.shape
        probs = torch.empty_like(logits)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        correct = torch.empty((), device=logits.device, dtype=torch.int32)
        _hip.check(_hip.lib().air_softmax_ce_fwd(dptr(logits), dptr(labels, torch.int64), ci(B), ci(C), dptr(probs),
                                                 dptr(loss), dptr(correct, torch.int32), stream()),
                   "air_softmax_ce_fwd")
        ctx.save_for_backward(probs, labels)
        owner.last_correct = correct
        return loss

    @staticmethod
    def backward(ctx, g):
        probs, labels = ctx.saved_tensors
        B, C = probs.shape
        d = torch.empty_like(probs)
        _hip.check(_hip.lib().air_softmax_ce_bwd(dptr(probs), dptr(labels, torch.int64), ci(B), ci(C),
                                                 dptr(g.reshape(1).float().contiguous()), dptr(d), stream()),
                   "air_softmax_ce_bwd")
        return d, None, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() with default arguments (main_train.py:251): mean over the batch.
    ``last_correct`` holds #(argmax == label) of the last call as a GPU scalar (the accuracy
    counters of main_train.py:383-385 without a second pass over the logits)."""

    def __init__(self):
        super().__init__()
        self.last_correct = None

    def forward(self, logits, labels):
        if not logits.is_cuda:
            raise _hip.AirError("CrossEntropyLoss HIP path needs GPU tensors; there is no CPU fallback")
        if logits.dim() != 2:
            raise ValueError("expected (B, C) logits")
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        return _CEFn.apply(logits.float(), labels, self)


class TensorAdam:
    """torch.optim.Adam(module.parameters(), lr, betas, eps, weight_decay) as configured for the
    classifiers at main_train.py:215-216 (lr_d = 1e-4, coupled L2 5e-4): one fused launch per tensor."""

    def __init__(self, module, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4):
        self.params = list(module.parameters())
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self.step_count = 0
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.step_count += 1
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(id(p))
            if st is None:
                st = self.state[id(p)] = (torch.zeros_like(p.data).view(-1), torch.zeros_like(p.data).view(-1))
            ops.adam_step(p.data.view(-1), p.grad.contiguous().view(-1), st[0], st[1], self.step_count, g["lr"],
                          g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], grad_scale)


class AdversarialTrainer(Trainer):
    """The ``--ADV_AUG`` train step (main_train.py:375-409 + :420-453).

    ``n_channels``: int (LA_aug / DF_aug: one classifier over ``channels (B,)``) or a tuple
    (LAPA_aug / DFPA_aug: codec and device classifiers over ``channels (B, 2)``).
    ``recompute=True`` is the reference's behaviour: after the encoder update the batch goes
    through the encoder AGAIN (train mode: BatchNorm statistics are updated a second time) to
    train the classifiers on detached features.  ``recompute=False`` trains them on the detached
    features of the first forward (one encoder forward per step instead of two: a deliberate,
    documented deviation)."""

    def __init__(self, model, n_channels, lambda_=0.05, lr_d=1e-4, recompute=True, **kw):
        super().__init__(model, **kw)
        counts = (n_channels,) if isinstance(n_channels, int) else tuple(n_channels)
        enc_dim = self.loss.feat_dim
        self.classifiers = [ChannelClassifier(enc_dim, n, lambda_).to(self.device) for n in counts]
        if self.world > 1:  # same classifier replicas everywhere (the encoder is synchronised by Trainer)
            for c in self.classifiers:
                for t in list(c.parameters()) + list(c.buffers()):
                    td.broadcast(t.data, src=0)
        self.classifier_optimizers = [TensorAdam(c, lr=lr_d) for c in self.classifiers]
        self.criterion = CrossEntropyLoss()
        self.lr_d = lr_d
        self.recompute = recompute
        self.last = {}

    def set_epoch(self, epoch_num, lr_decay=0.5, interval=30):
        super().set_epoch(epoch_num, lr_decay, interval)
        for opt in self.classifier_optimizers:  # main_train.py:301-306
            adjust_learning_rate(self.lr_d, opt, epoch_num, lr_decay, interval)

    def _targets(self, channels):
        channels = channels.to(self.device)
        if len(self.classifiers) == 1:
            return [channels.reshape(-1)]
        return [channels[:, i].contiguous() for i in range(len(self.classifiers))]

    def step_features(self, feat, labels, channels=None, epoch_num=1):
        if channels is None:
            return super().step_features(feat, labels)
        targets = self._targets(channels)
        self.model.train()
        for c in self.classifiers:
            c.train()
        self.feat_optimizer.zero_grad()
        self.loss_optimizer.zero_grad()
        feats, _ = self.model(feat)
        loss, neg_scores = self.loss(feats, labels)
        feat_loss = loss * self.weight_loss
        adv = None
        if epoch_num > 0:  # main_train.py:377
            for c, tgt in zip(self.classifiers, targets):
                l = self.criterion(c(feats), tgt)
                adv = l if adv is None else adv + l
            feat_loss = feat_loss + adv
        feat_loss.backward()
        scale = 1.0
        if self.world > 1:  # encoder + centre gradients are averaged over ranks
            air_dist.allreduce_grads(self.model, self.loss)
            scale = 1.0 / self.world
        self.feat_optimizer.step(grad_scale=scale)
        self.loss_optimizer.step(grad_scale=scale)
        # phase 2 (main_train.py:420-453): train the classifiers on detached features
        if self.recompute:
            with torch.no_grad():
                feats2 = self.model(feat)[0]
        else:
            feats2 = feats.detach()
        closs = []
        for c, opt, tgt in zip(self.classifiers, self.classifier_optimizers, targets):
            lc = self.criterion(c(feats2.detach()), tgt)
            opt.zero_grad()
            lc.backward()
            if self.world > 1:
                # the classifiers are replicas too: without this every rank would train its own, and the
                # gradient-reversal term each rank feeds its encoder would drift apart
                works = [td.all_reduce(p.grad, op=td.ReduceOp.SUM, async_op=True)
                         for p in c.parameters() if p.grad is not None]
                for w in works:
                    w.wait()
            opt.step(grad_scale=scale)
            closs.append(lc.detach())
        self.last = {"adv_loss": None if adv is None else adv.detach(), "classifier_loss": closs}
        return loss.detach(), neg_scores

    def step(self, pcm, labels, channels=None, start=None, epoch_num=1):
        if self.augment is not None:
            pcm = self.augment(pcm)
        return self.step_features(self.features(pcm, start), labels, channels, epoch_num)
