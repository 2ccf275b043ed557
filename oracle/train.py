"""Oracle (test infrastructure): one train step of the hot path.

Restates the inner loop of main_train.py:310-409 for ``--add_loss ang_iso``:
forward -> AngularIsoLoss -> backward -> Adam(model) + SGD(centre), with the
step-decay LR of main_train.py:144-147.
"""
import numpy as np
import torch

from . import ecapa as ecapa_oracle
from . import resnet as resnet_oracle
from .loss import ocsoftmax_forward


def lr_at_epoch(lr0, epoch, lr_decay=0.5, interval=30):
    """main_train.py:144-147."""
    return lr0 * (lr_decay ** (epoch // interval))


def adam_step_(p, g, m, v, step, lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4):
    """torch.optim.Adam as configured at main_train.py:175-176 (coupled L2 decay,
    no amsgrad).  In place on torch tensors; ``step`` is the 1-based count."""
    g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / np.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def sgd_step_(p, g, lr):
    """torch.optim.SGD(lr) as at main_train.py:272 (no momentum / decay)."""
    p.add_(g, alpha=-lr)


class OracleTrainer:
    """Functional train loop over a parameter dict (CPU)."""

    def __init__(self, model, params, center, lr=5e-4, r_real=0.9, r_fake=0.2, alpha=20.0,
                 weight_loss=1.0, bf16=False, **ecapa_options):
        assert model in ("resnet", "ecapa")
        self.model = model
        self.bf16 = bf16  # ECAPA only: BASELINE configs[2] arithmetic (oracle/ecapa.py)
        self.ecapa_options = ecapa_options  # ECAPA only: context= / summed= (ecapa_tdnn.py:99)
        self.params = {k: v.clone() for k, v in params.items()}
        self.center = center.clone()
        self.lr = lr
        self.r_real, self.r_fake, self.alpha = r_real, r_fake, alpha
        self.weight_loss = weight_loss
        self.step_count = 0
        self.m = {}
        self.v = {}
        self.relu = None  # ResNet only: a resnet_oracle.ReluProbe standing in for F.relu (ReLU-decision parity)

    def trainable(self):
        return [k for k, v in self.params.items()
                if v.dtype.is_floating_point and not resnet_oracle.is_buffer(k)]

    def forward(self, x, training=True, noise=None, updates=None, taps=None):
        if self.model == "resnet":
            return resnet_oracle.resnet18_forward(self.params, x, training, noise, updates, taps, relu=self.relu)
        opts = {k: v for k, v in self.ecapa_options.items() if k != "encoder_type"}  # (ASP is the shape of attention.3)
        return ecapa_oracle.ecapa_forward(self.params, x, training=training, updates=updates, taps=taps,
                                          bf16=self.bf16, **opts)

    def loss_and_grads(self, x, labels, noise=None):
        names = self.trainable()
        for k in names:
            self.params[k] = self.params[k].detach().requires_grad_(True)
        center = self.center.detach().requires_grad_(True)
        updates = {}
        feat, _ = self.forward(x, True, noise, updates)
        loss, neg_scores = ocsoftmax_forward(feat, center, labels, self.r_real, self.r_fake, self.alpha)
        (loss * self.weight_loss).backward()  # main_train.py:376, :406
        grads = {k: self.params[k].grad for k in names}
        gcenter = center.grad
        for k in names:
            self.params[k] = self.params[k].detach()
        return loss.detach(), neg_scores.detach(), feat.detach(), grads, gcenter, updates

    def step(self, x, labels, noise=None, epoch=0):
        loss, neg_scores, feat, grads, gcenter, updates = self.loss_and_grads(x, labels, noise)
        lr = lr_at_epoch(self.lr, epoch)
        self.step_count += 1
        with torch.no_grad():
            for k, g in grads.items():
                if g is None:  # fc_mu.* (ResNet) / fc7.*, bn7.* (ECAPA) get no grad under ang_iso
                    continue
                if k not in self.m:
                    self.m[k] = torch.zeros_like(self.params[k])
                    self.v[k] = torch.zeros_like(self.params[k])
                adam_step_(self.params[k], g, self.m[k], self.v[k], self.step_count, lr)
            sgd_step_(self.center, gcenter, lr)
            for k, v in updates.items():
                self.params[k] = v
            for k in list(self.params):
                if k.endswith("num_batches_tracked") and k.rsplit(".", 1)[0] + ".running_mean" in updates:
                    self.params[k] = self.params[k] + 1
        return loss, neg_scores, feat, grads, gcenter


def bf16_gradient_band(x, labels, got, mode=True):
    """Test helper for ECAPA in bf16 (oracle/ecapa.py, ``bf16=mode``: True = compute only, "resident").  bf16 rounding is
    discontinuous, so two correct evaluations of the same graph that differ in fp32 summation order
    disagree on gradients far more than in fp32.  Returns (band, errs): ``band`` = the oracle's own
    fp32-vs-fp64 relative-L2 gradient spread on this input (max / median over tensors) and the fp64 loss;
    ``errs[name]`` = (relative L2, cosine) of ``got[name]`` (flat float64 arrays) against the fp64 evaluation."""
    from .filler import fill_state, fill_value
    shapes = ecapa_oracle.ecapa_shapes()
    p32 = fill_state(shapes)
    p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in p32.items()}
    t64 = OracleTrainer("ecapa", p64, fill_value("center", (1, 256)).double(), bf16=mode)
    l64, _, _, g64, _, _ = t64.loss_and_grads(x.double(), labels)
    t32 = OracleTrainer("ecapa", p32, fill_value("center", (1, 256)), bf16=mode)
    _, _, _, g32, _, _ = t32.loss_and_grads(x, labels)
    own, own_cos, errs = [], [], {}
    for k, ref in g64.items():
        if ref is None or k in ("attention.2.bias", "attention.3.bias"):  # analytically zero gradients
            continue
        r = ref.numpy().ravel()
        nr = np.linalg.norm(r) + 1e-30
        o32 = g32[k].double().numpy().ravel()
        own.append(np.linalg.norm(o32 - r) / nr)
        own_cos.append(float(o32 @ r) / (np.linalg.norm(o32) * nr + 1e-30))
        g = got[k]
        errs[k] = (np.linalg.norm(g - r) / nr, float(g @ r) / (np.linalg.norm(g) * nr + 1e-30))
    # "min_cos": the smallest cosine the oracle's own fp32 evaluation reaches against its fp64 one
    return {"max": float(max(own)), "median": float(np.median(own)), "loss64": l64.item(),
            "min_cos": float(min(own_cos))}, errs
