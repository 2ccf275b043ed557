import numpy as np, torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import train as o_train
from oracle.filler import fill_module_, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Res2Net2, Bottle2neck
from asvspoof2021_air_amd.loss import AngularIsoLoss
torch.set_num_threads(8)
B, T = 32, 96
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
fill_module_(m)
m = m.cuda().train().set_compute_dtype("bf16")
lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
fill_module_(lossm); lossm = lossm.cuda()
x = synth_feat((B, 60, T), seed=400 + T)
labels = (torch.arange(B) % 3 != 0).long()
feat, _ = m(x.cuda()); loss, _ = lossm(feat, labels.cuda()); loss.backward()
got = {k: p.grad.cpu().double().numpy().ravel() for k, p in m.named_parameters() if p.grad is not None}
band, errs = o_train.bf16_gradient_band(x, labels, got, "resident")
print({k: v for k, v in band.items() if not hasattr(v, "shape") or v.size < 4})
for k, (e, c) in sorted(errs.items(), key=lambda kv: -kv[1][0])[:12]:
    print("%-28s err %.3f cos %.4f" % (k, e, c))
