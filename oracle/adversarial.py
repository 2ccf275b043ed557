"""Oracle (test infrastructure): the adversarial channel-classifier branch.

Follows model.py:976-1018 and main_train.py:211-224,377-403,420-453 (PyTorch-CPU):

* ``GradientReversalFunction`` (model.py:976-995): identity forward, ``dx = -lambda * dy``.
* ``ChannelClassifier.forward`` (model.py:1007-1023): GRL -> Linear(enc, enc//2) -> Dropout(0.3)
  -> ReLU -> Linear(enc//2, nclasses) -> ReLU.  The dropout keep-mask (already scaled by
  1/(1-p)) is an explicit argument so runs are reproducible; ``None`` = eval mode.
* ``nn.CrossEntropyLoss()`` (main_train.py:251): mean over the batch of -log softmax[label].
Pinned against the real reference modules by tests/golden/make_golden_adv.py -> adv.npz.
"""
import torch
import torch.nn.functional as F


def classifier_shapes(enc_dim=256, nclasses=10):
    return {"classifier.0.weight": (enc_dim // 2, enc_dim), "classifier.0.bias": (enc_dim // 2,),
            "classifier.3.weight": (nclasses, enc_dim // 2), "classifier.3.bias": (nclasses,)}


class _GRL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lambda_):
        ctx.lambda_ = lambda_
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return -ctx.lambda_ * g, None


def classifier_forward(p, feats, lambda_, keep=None):
    x = _GRL.apply(feats, lambda_)
    h = F.linear(x, p["classifier.0.weight"], p["classifier.0.bias"])
    if keep is not None:
        h = h * keep
    h = F.relu(h)
    return F.relu(F.linear(h, p["classifier.3.weight"], p["classifier.3.bias"]))


def cross_entropy(logits, labels):
    return F.cross_entropy(logits, labels)


def loss_and_grads(p, feats, labels, lambda_, keep=None):
    """Returns (loss, logits, dfeats, grads dict)."""
    p = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    f = feats.detach().clone().requires_grad_(True)
    logits = classifier_forward(p, f, lambda_, keep)
    loss = cross_entropy(logits, labels)
    loss.backward()
    return loss.detach(), logits.detach(), f.grad, {k: v.grad for k, v in p.items()}
