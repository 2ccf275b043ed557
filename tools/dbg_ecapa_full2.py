import sys, torch, numpy as np
from oracle.filler import fill_module_, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd import ops
B, T = int(sys.argv[1]), int(sys.argv[2])
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
x = synth_feat((B,60,T), seed=3).cuda()
feat, out, S = m._forward_impl(x, save=True)
o_cs, o_wg, o_dg = ops.channel_sum, ops.conv1d_wgrad, ops.conv1d_dgrad
def cs(xx, out=None):
    r = o_cs(xx, out=out); torch.cuda.synchronize()
    ref = xx.double().sum((0,2)); e = float((r.double()-ref).abs().max()/(ref.abs().max()+1e-30))
    if e > 1e-4: print('channel_sum BAD', tuple(xx.shape), xx.stride(), e)
    return r
def wg(xx, dy, wshape, dil=1, pad=0, out=None):
    r = o_wg(xx, dy, wshape, dil, pad, out=out); torch.cuda.synchronize()
    if wshape[2] == 1:
        ref = torch.einsum('bot,bit->oi', dy.double(), xx.double()).unsqueeze(2)
        e = float((r.double()-ref).abs().max()/(ref.abs().max()+1e-30))
        if e > 1e-4: print('wgrad BAD', tuple(xx.shape), xx.stride(), tuple(dy.shape), dy.stride(), wshape, e)
    return r
ops.channel_sum, ops.conv1d_wgrad = cs, wg
dfeat = synth_feat((B,256), 10).cuda()*0.01
m._backward_impl(S, dfeat, None)
print('done')
