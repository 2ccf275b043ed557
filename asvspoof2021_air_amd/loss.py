"""Drop-in for the reference's ``loss.AngularIsoLoss`` / ``loss.OCSoftmax``
(loss.py:62-97, :176-206): the OC-Softmax (``ang_iso``) head.

Same constructor defaults, ``forward(x, labels) -> (loss, -scores)``, attribute
``center`` (1, feat_dim) read by main_train.py:610.  Forward and backward are
one HIP launch each (csrc/ocsoftmax.hip).
"""
import torch
import torch.nn as nn

from . import _hip, ops


class _OCSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, center, labels, r_real, r_fake, alpha):
        x = x.contiguous()
        loss, neg = ops.ocsoftmax_fwd(x, center.detach().contiguous(), labels, r_real, r_fake, alpha)
        ctx.save_for_backward(x, center, labels)
        ctx.cfg = (r_real, r_fake, alpha)
        ctx.mark_non_differentiable(neg)
        return loss, neg

    @staticmethod
    def backward(ctx, dloss, _dneg):
        x, center, labels = ctx.saved_tensors
        r_real, r_fake, alpha = ctx.cfg
        g = dloss.reshape(1).float().contiguous()
        dx, dc = ops.ocsoftmax_bwd(x, center.detach().contiguous(), labels, r_real, r_fake, alpha, gscale=g)
        return dx, dc, None, None, None, None


class AngularIsoLoss(nn.Module):
    def __init__(self, feat_dim=2, r_real=0.9, r_fake=0.5, alpha=20.0):
        super().__init__()
        self.feat_dim = feat_dim
        self.r_real = r_real
        self.r_fake = r_fake
        self.alpha = alpha
        self.center = nn.Parameter(torch.randn(1, self.feat_dim))
        nn.init.kaiming_uniform_(self.center, 0.25)
        self.softplus = nn.Softplus()

    def forward(self, x, labels):
        """x: (B, feat_dim) GPU features; labels: (B,) 0 = bona fide, 1 = spoof."""
        if not x.is_cuda:
            raise _hip.AirError("OC-Softmax HIP path needs GPU tensors; there is no CPU fallback")
        if x.dim() != 2 or x.shape[1] != self.feat_dim:
            raise ValueError("expected (B, %d) features, got %s" % (self.feat_dim, tuple(x.shape)))
        labels = labels.to(device=x.device, dtype=torch.int64).contiguous()
        return _OCSoftmaxFn.apply(x.float(), self.center, labels, float(self.r_real),
                                  float(self.r_fake), float(self.alpha))


class OCSoftmax(AngularIsoLoss):
    """loss.py:176-206: identical arithmetic, kept under its second name."""
