import torch, numpy as np
from oracle import ecapa as o_ecapa
from oracle.filler import fill_module_, fill_state, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
B,T = 2,750
x = synth_feat((B,60,T), seed=400+T)
p32 = fill_state(o_ecapa.ecapa_shapes())
p64 = {k:(v.double() if v.dtype.is_floating_point else v) for k,v in p32.items()}
taps = {}
fo, oo = o_ecapa.ecapa_forward(p64, x.double(), training=True, taps=taps)
feat, out, S = m._forward_impl(x.cuda(), save=True)
def rel(a,b):
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    return float((a-b).abs().max()/b.abs().max())
c = S['cat123']
print('x1', rel(c[:, :512], taps['x1']), 'x2', rel(c[:,512:1024], taps['x2']), 'x3', rel(c[:,1024:], taps['x3']))
print('layer4', rel(S['x4'], taps['layer4']))
print('w', rel(S['wts'], taps['w']))
print('mu', rel(S['pooled'][:, :1536], taps['mu']), 'sg', rel(S['pooled'][:,1536:], taps['sg']))
print('feat', rel(feat, fo), 'out', rel(out, oo))
# f32 oracle for comparison
t32={}
f32,o32 = o_ecapa.ecapa_forward(p32, x, training=True, taps=t32)
print('oracle32: x1', rel(t32['x1'],taps['x1']), 'x3', rel(t32['x3'],taps['x3']), 'layer4', rel(t32['layer4'],taps['layer4']), 'w', rel(t32['w'],taps['w']), 'mu', rel(t32['mu'],taps['mu']), 'sg', rel(t32['sg'],taps['sg']), 'feat', rel(f32,fo))
