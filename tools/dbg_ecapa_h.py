"""Debug: ECAPA bf16-resident path against the fp32 and the bf16-compute paths of the same library (same weights,
same batch): feature / loss / per-tensor gradient agreement."""
import sys, numpy as np, torch
from oracle.filler import fill_module_, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd.loss import AngularIsoLoss
B, T = int(sys.argv[1]), int(sys.argv[2])
x = synth_feat((B, 60, T), seed=750).cuda()
labels = (torch.arange(B) % 3 != 0).long().cuda()
res = {}
for dt in ("fp32", "bf16c", "bf16"):
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    fill_module_(m)
    m = m.cuda().train().set_compute_dtype(dt)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    lossm = lossm.cuda()
    feat, _ = m(x)
    loss, _ = lossm(feat, labels)
    loss.backward()
    torch.cuda.synchronize()
    res[dt] = (feat.detach().cpu().double(), loss.item(), {k: p.grad.detach().cpu().double().ravel() for k, p in m.named_parameters() if p.grad is not None})
    print(dt, "loss %.6f" % loss.item(), "finite", bool(torch.isfinite(feat).all()))
f0, l0, g0 = res["fp32"]
for dt in ("bf16c", "bf16"):
    f, l, g = res[dt]
    rel = [(np.linalg.norm(g[k] - g0[k]) / (np.linalg.norm(g0[k]) + 1e-30), k) for k in g0]
    cos = [float(g[k] @ g0[k]) / (np.linalg.norm(g[k]) * np.linalg.norm(g0[k]) + 1e-30) for k in g0]
    rel.sort()
    print("%s vs fp32: feat rel-L2 %.4f  loss rel %.2e  grads rel-L2 median %.3f  p90 %.3f  worst %s  cos median %.4f min %.4f" % (
        dt, float((f - f0).norm() / f0.norm()), abs(l / l0 - 1), rel[len(rel) // 2][0], rel[int(len(rel) * 0.9)][0], rel[-3:], float(np.median(cos)), min(cos)))
