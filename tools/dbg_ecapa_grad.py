import sys, torch, numpy as np
from oracle import ecapa as o_ecapa, train as o_train
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
B, T = int(sys.argv[1]), int(sys.argv[2]); use_out = int(sys.argv[3])
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
x = synth_feat((B,60,T), seed=3); wo, wf = synth_feat((B,2),9), synth_feat((B,256),10)
feat, out = m(x.float().cuda())
loss = (feat*wf.float().cuda()).sum()*0.01 + ((out*wo.float().cuda()).sum() if use_out else 0)
loss.backward()
DT = torch.float64 if len(sys.argv) > 4 else torch.float32
tr = o_train.OracleTrainer("ecapa", {k:(v.to(DT) if v.dtype.is_floating_point else v) for k,v in fill_state(o_ecapa.ecapa_shapes()).items()}, fill_value("center",(1,256)))
x = x.to(DT); wo = wo.to(DT); wf = wf.to(DT)
for k in tr.trainable(): tr.params[k] = tr.params[k].detach().requires_grad_(True)
fo, oo = tr.forward(x)
(((fo*wf).sum()*0.01) + ((oo*wo).sum() if use_out else 0)).backward()
print('feat err', float((feat.detach().cpu().to(DT)-fo.detach()).abs().max()), 'threads', torch.get_num_threads())
bad = []
for k,p in m.named_parameters():
    r = tr.params[k].grad
    if r is None or p.grad is None: continue
    e = float((p.grad.cpu().to(DT)-r).abs().max()/ (r.abs().max()+1e-30))
    if e > 1e-3: bad.append((k, e, float(r.abs().max())))
print(len(bad), "bad"); [print(b) for b in bad if b[0].startswith("layer3")]
