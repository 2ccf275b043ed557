"""Equal error rate as the trainer computes it (main_train.py:662-664 calls eval_metrics.compute_eer on
both score polarities).  Host numpy, like the reference: a few hundred to a few ten thousand scores.

Formulation: sort all trials by (score, class) with bona fide trials first among equal scores, then walk the
thresholds once - a threshold at the k-th sorted trial rejects the k lowest trials, so the false-rejection
rate is the share of targets among them and the false-acceptance rate the share of non-targets above them.
Equals the reference's curve (eval_metrics.py:19-46) point for point, ties included
(tests/golden/eer.npz, 1e-12)."""
import numpy as np


def compute_det_curve(target_scores, nontarget_scores):
    """(frr, far, thresholds), each of length n_trials + 1; entry 0 = nothing rejected."""
    tar = np.asarray(target_scores, dtype=np.float64).ravel()
    non = np.asarray(nontarget_scores, dtype=np.float64).ravel()
    scores = np.concatenate((tar, non))
    is_non = np.zeros(scores.size, dtype=bool)
    is_non[tar.size:] = True
    order = np.lexsort((is_non, scores))  # by score; among ties targets first
    rejected_non = np.cumsum(is_non[order])
    rejected_tar = np.arange(1, scores.size + 1) - rejected_non
    frr = np.empty(scores.size + 1)
    far = np.empty(scores.size + 1)
    frr[0], far[0] = 0.0, 1.0
    frr[1:] = rejected_tar / tar.size
    far[1:] = (non.size - rejected_non) / non.size
    thresholds = np.empty(scores.size + 1)
    thresholds[1:] = scores[order]
    thresholds[0] = thresholds[1] - 0.001
    return frr, far, thresholds


def compute_eer(target_scores, nontarget_scores):
    """(eer, threshold) at the curve point where |FRR - FAR| is smallest (first such point)."""
    frr, far, thresholds = compute_det_curve(target_scores, nontarget_scores)
    k = int(np.argmin(np.abs(frr - far)))
    return float((frr[k] + far[k]) / 2.0), float(thresholds[k])


def eer_both_polarities(scores, labels):
    """min over both score polarities with label 0 = bona fide (main_train.py:662-664)."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    return min(compute_eer(scores[labels == 0], scores[labels == 1])[0],
               compute_eer(-scores[labels == 0], -scores[labels == 1])[0])
