"""Oracle (test infrastructure): OC-Softmax (``ang_iso``) head.

``AngularIsoLoss.forward`` (loss.py:73-97) and ``OCSoftmax.forward``
(loss.py:187-206) are the same arithmetic.
"""
import numpy as np
import torch
import torch.nn.functional as F


def ocsoftmax_forward(x, center, labels, r_real=0.9, r_fake=0.5, alpha=20.0):
    """loss.py:79-97.  x: (B, D); center: (1, D); labels: (B,) {0 bona fide, 1 spoof}.
    Returns (loss, -scores) where scores = cos(x, center)."""
    w = F.normalize(center, p=2, dim=1)  # :79, eps 1e-12
    xn = F.normalize(x, p=2, dim=1)  # :80
    scores = (xn @ w.t()).squeeze(1)  # :82
    margin = torch.where(labels == 0, r_real - scores, scores - r_fake)  # :85-86
    loss = F.softplus(alpha * margin).mean()  # :93 (nn.Softplus beta=1, threshold=20)
    return loss, -scores


def ocsoftmax_grads_f64(x, center, labels, r_real=0.9, r_fake=0.5, alpha=20.0):
    """Closed-form float64 gradients (independent of autograd) of the mean
    softplus loss w.r.t. x and center; used to cross-check the HIP backward."""
    x = np.asarray(x, np.float64)
    c = np.asarray(center, np.float64).reshape(-1)
    lab = np.asarray(labels)
    nx = np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    nc = max(np.linalg.norm(c), 1e-12)
    xh, wh = x / nx, c / nc
    s = xh @ wh
    sign = np.where(lab == 0, -1.0, 1.0)
    m = np.where(lab == 0, r_real - s, s - r_fake)
    z = alpha * m
    sig = np.where(z > 20.0, 1.0, 1.0 / (1.0 + np.exp(-z)))  # softplus' with torch threshold
    dl_ds = sig * alpha * sign / x.shape[0]
    # ds/dx = (wh - s xh)/|x| ; ds/dc = (xh - s wh)/|c|
    gx = dl_ds[:, None] * (wh[None, :] - s[:, None] * xh) / nx
    gc = (dl_ds[:, None] * (xh - s[:, None] * wh[None, :])).sum(0) / nc
    loss = np.where(z > 20.0, z, np.log1p(np.exp(np.minimum(z, 20.0)))).mean()
    return loss, -s, gx, gc.reshape(1, -1)
