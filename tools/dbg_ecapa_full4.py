import sys, torch, numpy as np, torch.nn.functional as F
from oracle import ecapa as oe
from oracle.filler import fill_module_, fill_state, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd import ops
B, T = int(sys.argv[1]), int(sys.argv[2])
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
x = synth_feat((B,60,T), seed=3)
feat, out, S = m._forward_impl(x.cuda(), save=True)
calls = []
o_bn = ops.bn_bwd
def bn(xx, dy, *a, **k):
    calls.append((tuple(xx.shape), dy.clone()))
    r = o_bn(xx, dy, *a, **k); torch.cuda.synchronize(); calls[-1] += (r[0].clone(),)
    return r
ops.bn_bwd = bn
dfeat = synth_feat((B,256), 10)*0.01
m._backward_impl(S, dfeat.cuda(), None)
# oracle with retained grads on layer3 internals
p = {k: (v.double().requires_grad_(True) if (v.dtype.is_floating_point and not k.split('.')[-1].startswith('running')) else (v.double() if v.dtype.is_floating_point else v)) for k,v in fill_state(oe.ecapa_shapes()).items()}
keep = {}
orig_bn = oe._bn
def bn_hook(xx, pp, prefix, training, updates):
    y = orig_bn(xx, pp, prefix, training, updates)
    if prefix in ("layer3.bn1", "layer3.bn3", "layer2.bn1"):
        xx.retain_grad(); y.retain_grad(); keep[prefix] = (xx, y)
    return y
oe._bn = bn_hook
fo, oo = oe.ecapa_forward(p, x.double(), training=True)
(fo*dfeat.double()).sum().backward()
def rel(a,b): return float((a.detach().cpu().double()-b).abs().max()/(b.abs().max()+1e-30))
# my calls in order: bn5, attention.2, then layer3: se bn (B,128,1), bn3 (B,512,T), 7x bns, bn1 (B,512,T) ...
big = [c for c in calls if c[0]==(B,512,T)]
print('n big bn calls', len(big))
# order: layer3.bn3, layer3.bn1, layer2.bn3, layer2.bn1, ...
print('layer3.bn3: dy', rel(big[0][1], keep["layer3.bn3"][1].grad), 'dx', rel(big[0][2], keep["layer3.bn3"][0].grad*(keep["layer3.bn3"][0]>0)))
print('layer3.bn1: dy', rel(big[1][1], keep["layer3.bn1"][1].grad), 'dx', rel(big[1][2], keep["layer3.bn1"][0].grad*(keep["layer3.bn1"][0]>0)))
d = (big[1][1].cpu().double()-keep["layer3.bn1"][1].grad).abs().amax((0,2))
print('per-channel-group max err of do1:', [float(d[i*64:(i+1)*64].max()) for i in range(8)], 'scale', float(keep["layer3.bn1"][1].grad.abs().max()))
G = m.arena().grad_views()
for k in ("layer3.conv1.bias","layer3.conv1.weight","layer3.bn1.weight","layer2.conv1.bias","conv1.weight"):
    print(k, rel(G[k], p[k].grad))
