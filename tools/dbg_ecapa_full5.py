import sys, torch
from oracle.filler import fill_module_, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
B, T = int(sys.argv[1]), int(sys.argv[2])
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
x = synth_feat((B,60,T), seed=3).cuda(); wf = synth_feat((B,256),10).cuda()
feat, out, S = m._forward_impl(x, save=True)
m._backward_impl(S, (wf*0.01).contiguous(), None); torch.cuda.synchronize()
G1 = {k: v.clone() for k, v in m.arena().grad_views().items()}
fill_module_(m)
feat2, out2 = m(x)
print('fwd same', float((feat-feat2).abs().max()))
((feat2*wf).sum()*0.01).backward(); torch.cuda.synchronize()
for k,p in m.named_parameters():
    if p.grad is None: continue
    e = float((p.grad-G1[k]).abs().max()/(G1[k].abs().max()+1e-30))
    if e > 1e-5: print('DIFF', k, e, p.grad.data_ptr()==m.arena().grad_view(k).data_ptr())
print('done')
