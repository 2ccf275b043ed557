"""The HIP path's own run-to-run spread on the EER fixtures: the same recipe from initial weights with one element moved
by 1e-7 (as make_golden_eer3.py `spread` does for the reference).  Usage: dbg_eer_spread.py <resnet|ecapa> <dtype> [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_eer_gpu as t
which, dtype = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
g = np.load(os.path.join(ROOT, "tests", "golden", "synth_eer3_%s.npz" % which))
for k in range(n):
    tr, model, lossm, epoch_loss, scores, eer, lab_ho, pcm_ho = t._run(g, which, dtype, perturb=k)
    print(which, dtype, "perturb", k, "EER %.5f" % eer, "errors", t._error_counts(scores, lab_ho), "final loss %.4f" % epoch_loss[-1], flush=True)
