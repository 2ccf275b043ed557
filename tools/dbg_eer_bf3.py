"""Is the slower 4 s ResNet loss curve under the split-bf16 kernels chaos or bias?  Final-epoch loss and wrong trials of
the unperturbed run + 5 perturbed runs (one initial weight moved by 1e-7) under CONV_S2 = 3 (all f32) and 31 (split-bf16)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa
import test_eer_gpu as T
from asvspoof2021_air_amd import _hip
g = conftest.load_golden("synth_eer4s_resnet.npz")
for opt in (3, 31):
    floors, errs, mids = [], [], []
    with _hip.options(CONV_S2=opt):
        for k in range(int(os.environ.get("NRUN", "6"))):
            r = T._run(g, "resnet", "fp32", perturb=k) if k else T._run(g, "resnet", "fp32")
            floors.append(float(r[3][-1])); mids.append(float(r[3][6])); errs.append(sum(T._error_counts(r[4], r[6])))
    print("CONV_S2=%d final losses %s median %.4f | epoch-7 %s | wrong trials %s" % (
        opt, np.round(floors, 4).tolist(), np.median(floors), np.round(mids, 3).tolist(), errs), flush=True)
print("reference final", float(g["epoch_loss"][-1]), "epoch-7", float(g["epoch_loss"][6]))
