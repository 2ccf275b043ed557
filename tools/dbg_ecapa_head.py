import sys, torch, numpy as np, torch.nn.functional as F
torch.set_num_threads(8)
from oracle import ecapa as oe
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat
from oracle.loss import ocsoftmax_forward
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd import ops
B,T = 8,64
m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); fill_module_(m); m = m.cuda().train()
x = synth_feat((B,60,T), seed=400+T)
feat, out, S = m._forward_impl(x.cuda(), save=True)
cap = {}
o_mask, o_asp, o_rsb, o_dg = ops.relu_mask_, ops.asp_bwd, ops.row_stats_bwd, ops.conv1d_dgrad
def mask(dx, y): cap['dx4'] = dx.clone(); return o_mask(dx, y)
def asp(xx, w, outp, dout, dx, accumulate=False):
    r = o_asp(xx, w, outp, dout, dx, accumulate); cap['dx_asp'] = dx.clone(); cap['dlogits'] = w.clone(); return r
def rsb(xx, mean, std, dmean, dstd, dx, accumulate=True, clamp_min=1e-4):
    if std is not None and xx.shape[1]==1536: cap['dmean']=dmean.clone(); cap['dstd']=dstd.clone(); cap['dx_before_ctx']=dx.clone()
    return o_rsb(xx, mean, std, dmean, dstd, dx, accumulate, clamp_min)
ops.relu_mask_, ops.asp_bwd, ops.row_stats_bwd = mask, asp, rsb
labels = (torch.arange(B)%3!=0).long()
fq = feat.detach().cpu().double().requires_grad_(True)
lq,_ = ocsoftmax_forward(fq, fill_value("center",(1,256)).double(), labels, 0.9, 0.2, 20.0); lq.backward()
dfeat = fq.grad.float()
print('dfeat max', float(dfeat.abs().max()))
m._backward_impl(S, dfeat.cuda(), None)
p = {k: (v.double().requires_grad_(True) if (v.dtype.is_floating_point and not k.split('.')[-1].startswith('running')) else (v.double() if v.dtype.is_floating_point else v)) for k,v in fill_state(oe.ecapa_shapes()).items()}
taps = {}
# replicate head of oracle forward with retained grads
def fwd():
    xx = x.double()
    h = oe._bn(F.relu(oe._conv(xx, p, "conv1", 1, 2)), p, "bn1", True, None)
    x1 = oe.bottle2neck(h, p, "layer1", 2, 8, True, None); x2 = oe.bottle2neck(x1, p, "layer2", 3, 8, True, None); x3 = oe.bottle2neck(x2, p, "layer3", 4, 8, True, None)
    x4 = F.relu(oe._conv(torch.cat((x1,x2,x3),1), p, "layer4")); x4.retain_grad(); taps['x4']=x4
    t = x4.shape[-1]
    mean = x4.mean(2, keepdim=True); std = torch.sqrt(x4.var(2, keepdim=True).clamp(min=1e-4)); mean.retain_grad(); std.retain_grad(); taps['mean']=mean; taps['std']=std
    gx = torch.cat((x4, mean.repeat(1,1,t), std.repeat(1,1,t)),1)
    a = oe._bn(F.relu(oe._conv(gx, p, "attention.0")), p, "attention.2", True, None)
    lg = oe._conv(a, p, "attention.3"); lg.retain_grad(); taps['lg']=lg
    w = torch.softmax(lg, dim=2)
    mu = torch.sum(x4*w, dim=2); sg = torch.sqrt((torch.sum((x4**2)*w, dim=2)-mu**2).clamp(min=1e-4))
    pooled = torch.cat((mu,sg),1); pooled.retain_grad(); taps['pooled']=pooled
    y = oe._bn(pooled, p, "bn5", True, None)
    return F.linear(y, p["fc6.weight"], p["fc6.bias"])
f = fwd(); (f*dfeat.double()).sum().backward()
def rel(a,b): return float((a.detach().cpu().double()-b).abs().max()/(b.abs().max()+1e-30))
print('feat', rel(feat, f))
print('dx4 total', rel(cap['dx4'], taps['x4'].grad), ' dlogits', rel(cap['dlogits'], taps['lg'].grad))
# decompose oracle: asp-only part of dx4
x4 = taps['x4'].detach(); lg = taps['lg'].detach()
x4a = x4.clone().requires_grad_(True); w = torch.softmax(lg, dim=2)
mu = torch.sum(x4a*w, dim=2); sg = torch.sqrt((torch.sum((x4a**2)*w, dim=2)-mu**2).clamp(min=1e-4))
torch.cat((mu,sg),1).backward(taps['pooled'].grad)
print('dx asp part', rel(cap['dx_asp'], x4a.grad))
vv = (torch.sum((x4**2)*w, dim=2)-mu.detach()**2)
G = m.arena().grad_views()
print('layer4.bias', rel(G['layer4.bias'], p['layer4.bias'].grad), 'layer4.weight', rel(G['layer4.weight'], p['layer4.weight'].grad), 'att3.w', rel(G['attention.3.weight'], p['attention.3.weight'].grad), 'bn5.w', rel(G['bn5.weight'], p['bn5.weight'].grad))
print('clamped fraction', float((vv<1e-4).double().mean()), 'near-threshold', float(((vv-1e-4).abs()<1e-6).double().mean()))
dx4 = cap['dx4'].cpu().double(); x4g = S['x4'].cpu().double()
mine_ref = (dx4*(x4g>0)).sum((0,2))
print('G vs torch-sum-of-my-dx4', rel(G['layer4.bias'], mine_ref))
orc = (taps['x4'].grad*(taps['x4']>0)).sum((0,2))
print('oracle recomputed vs param grad', rel(orc, p['layer4.bias'].grad))
print('mask mismatch count', int(((x4g>0)!=(taps['x4'].detach()>0)).sum()), 'of', x4g.numel())
print('x4 err', rel(S['x4'], taps['x4']))
d = (G['layer4.bias'].cpu().double()-p['layer4.bias'].grad).abs(); i = int(d.argmax())
print('worst channel', i, float(G['layer4.bias'][i]), float(p['layer4.bias'].grad[i]), 'max|grad|', float(p['layer4.bias'].grad.abs().max()))

print('dmean', rel(cap['dmean'], taps['mean'].grad.squeeze(2)), 'dstd', rel(cap['dstd'], taps['std'].grad.squeeze(2)))
print('|dmean| max', float(taps['mean'].grad.abs().max()), '|dx4| max', float(taps['x4'].grad.abs().max()), 'median |dx4|', float(taps['x4'].grad.abs().median()))
