import sys, torch, numpy as np
from oracle.filler import fill_module_, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd import ops
B, T = int(sys.argv[1]), int(sys.argv[2])
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
x = synth_feat((B,60,T), seed=3).cuda()
feat, out, S = m._forward_impl(x, save=True)
# monkeypatch _block_bwd to snapshot grads right after each block
snap = {}
orig = m._block_bwd
def patched(Sb, dout, G, pre):
    r = orig(Sb, dout, G, pre)
    torch.cuda.synchronize()
    for k in ("conv1.weight","conv1.bias","bn1.weight","conv3.weight"):
        snap[pre+k] = G[pre+k].clone()
    snap[pre+"dout"] = dout.clone()
    return r
m._block_bwd = patched
dfeat = synth_feat((B,256), 10).cuda()*0.01
grads = m._backward_impl(S, dfeat, None)
torch.cuda.synchronize()
G = m.arena().grad_views()
for k,v in snap.items():
    if k.endswith("dout"): continue
    print(k, 'changed after block bwd:', float((G[k]-v).abs().max()), 'max', float(v.abs().max()))
