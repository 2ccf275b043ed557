"""Oracle (test infrastructure): equal error rate.

Restates eval_metrics.py:19-46 (``compute_det_curve`` / ``compute_eer``).
"""
import numpy as np


def det_curve(target_scores, nontarget_scores):
    """eval_metrics.py:19-37: stable (merge) sort of the pooled scores, cumulative
    target counts -> FRR, remaining non-target counts -> FAR."""
    tgt = np.asarray(target_scores, dtype=np.float64)
    non = np.asarray(nontarget_scores, dtype=np.float64)
    n = tgt.size + non.size
    pooled = np.concatenate((tgt, non))
    is_tgt = np.concatenate((np.ones(tgt.size), np.zeros(non.size)))
    order = np.argsort(pooled, kind="mergesort")
    is_tgt = is_tgt[order]
    tgt_below = np.cumsum(is_tgt)
    non_above = non.size - (np.arange(1, n + 1) - tgt_below)
    frr = np.concatenate(([0.0], tgt_below / tgt.size))
    far = np.concatenate(([1.0], non_above / non.size))
    thr = np.concatenate(([pooled[order[0]] - 0.001], pooled[order]))
    return frr, far, thr


def compute_eer(target_scores, nontarget_scores):
    """eval_metrics.py:40-46: operating point minimising |FRR-FAR|; EER is the
    mean of the two rates there."""
    frr, far, thr = det_curve(target_scores, nontarget_scores)
    i = int(np.argmin(np.abs(frr - far)))
    return float((frr[i] + far[i]) / 2.0), float(thr[i])


def eer_both_polarities(scores, labels):
    """main_train.py:662-664 takes min over both score polarities.
    labels: 0 bona fide (target), 1 spoof."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    a = compute_eer(scores[labels == 0], scores[labels == 1])[0]
    b = compute_eer(-scores[labels == 0], -scores[labels == 1])[0]
    return min(a, b)
