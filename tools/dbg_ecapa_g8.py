import sys, torch, numpy as np
torch.set_num_threads(8)
from oracle import ecapa as o_ecapa, train as o_train
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd.loss import AngularIsoLoss
B,T = 8,64
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0); fill_module_(lossm); lossm = lossm.cuda()
x = synth_feat((B,60,T), seed=400+T); labels = (torch.arange(B)%3!=0).long()
feat, out = m(x.cuda()); loss,_ = lossm(feat, labels.cuda()); loss.backward()
P = fill_state(o_ecapa.ecapa_shapes())
t64 = o_train.OracleTrainer("ecapa", {k:(v.double() if v.dtype.is_floating_point else v) for k,v in P.items()}, fill_value("center",(1,256)).double())
lo, no, fo, g64, gc, _ = t64.loss_and_grads(x.double(), labels)
print('loss', loss.item(), lo.item(), 'feat err', float((feat.detach().cpu().double()-fo).abs().max()))
for k,p in m.named_parameters():
    if g64[k] is None or k in ("attention.2.bias","attention.3.bias"): continue
    e = float((p.grad.cpu().double()-g64[k]).abs().max()/(g64[k].abs().max()+1e-30))
    if e > 1e-4: print('%-28s %.3g' % (k, e))
