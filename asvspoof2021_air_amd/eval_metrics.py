"""Equal error rate as the trainer computes it (eval_metrics.py:19-46, main_train.py:662-664).
Host numpy, like the reference: scores are a few hundred floats per evaluation."""
import numpy as np


def compute_det_curve(target_scores, nontarget_scores):
    """eval_metrics.py:19-37."""
    target_scores = np.asarray(target_scores, dtype=np.float64)
    nontarget_scores = np.asarray(nontarget_scores, dtype=np.float64)
    n_scores = target_scores.size + nontarget_scores.size
    all_scores = np.concatenate((target_scores, nontarget_scores))
    labels = np.concatenate((np.ones(target_scores.size), np.zeros(nontarget_scores.size)))
    indices = np.argsort(all_scores, kind="mergesort")
    labels = labels[indices]
    tar_trial_sums = np.cumsum(labels)
    nontarget_trial_sums = nontarget_scores.size - (np.arange(1, n_scores + 1) - tar_trial_sums)
    frr = np.concatenate((np.atleast_1d(0), tar_trial_sums / target_scores.size))
    far = np.concatenate((np.atleast_1d(1), nontarget_trial_sums / nontarget_scores.size))
    thresholds = np.concatenate((np.atleast_1d(all_scores[indices[0]] - 0.001), all_scores[indices]))
    return frr, far, thresholds


def compute_eer(target_scores, nontarget_scores):
    """eval_metrics.py:40-46.  Returns (eer, threshold)."""
    frr, far, thresholds = compute_det_curve(target_scores, nontarget_scores)
    min_index = np.argmin(np.abs(frr - far))
    return float(np.mean((frr[min_index], far[min_index]))), float(thresholds[min_index])
