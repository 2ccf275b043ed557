import sys, torch, numpy as np, torch.nn.functional as F
from oracle import ecapa as o_ecapa
from oracle.filler import fill_module_, fill_state, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd import ops
B, T = int(sys.argv[1]), int(sys.argv[2])
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
inp = synth_feat((B,512,T), 1).cuda()
out = torch.empty(B,512,T, device='cuda')
blk = m.layer3
S = m._block_fwd(blk, inp, out, True, True)
dout = synth_feat((B,512,T), 2).cuda()
G = m.arena().grad_views()
dinp = m._block_bwd(S, dout, G, "layer3.")
# oracle block in fp64
p = {k: v.double() for k,v in fill_state(o_ecapa.ecapa_shapes()).items() if v.dtype.is_floating_point}
names = [k for k in p if k.startswith("layer3.") and not k.split('.')[-1].startswith('running')]
for k in names: p[k].requires_grad_(True)
xin = inp.cpu().double().requires_grad_(True)
yo = o_ecapa.bottle2neck(xin, p, "layer3", 4, 8, True, None)
yo.backward(dout.cpu().double())
def rel(a,b): return float((a.detach().cpu().double()-b).abs().max()/(b.abs().max()+1e-30))
print('out', rel(out, yo), 'dinp', rel(dinp, xin.grad))
for k in names:
    e = rel(G[k], p[k].grad)
    if e > 1e-3: print('BAD', k, e)
