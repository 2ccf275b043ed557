import torch, numpy as np
from oracle import resnet as o_resnet
from oracle.filler import fill_module_, fill_state, synth_feat
from asvspoof2021_air_amd.resnet import ResNet
from asvspoof2021_air_amd import ops
m = ResNet(3,256,'18',2); fill_module_(m); m = m.cuda().train()
B,T = 2,96
x = synth_feat((B,1,60,T), seed=200+T)
params = fill_state(o_resnet.resnet18_shapes())
taps = {}
torch.manual_seed(1234); noise = 1e-5*torch.randn(B,12,256)
fo, mo = o_resnet.resnet18_forward(params, x, True, noise, None, taps)
m.set_attention_noise(noise)
feat, mu, S = m._forward_impl(x.cuda(), None, save=True)
def rel(a,b): 
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    return float((a-b).abs().max()/b.abs().max())
print('conv1', rel(S['c1'], taps['conv1']))
outs = []
cur = None
for i,(blk, xin, stA, h, stB) in enumerate(S['blocks']):
    print('block',i,'input shape',tuple(xin.shape))
# compare layer outputs: block inputs of next layer
names = {2:'layer1',4:'layer2',6:'layer3'}
for i,(blk,xin,stA,h,stB) in enumerate(S['blocks']):
    if i in names: print(names[i], rel(xin, taps[names[i]]))
print('layer4', rel(S['l4'], taps['layer4']))
print('conv5', rel(S['c5'], taps['conv5']))
print('stats', rel(S['pooled'], taps['stats']))
print('feat', rel(feat, fo), 'mu', rel(mu, mo))
# first block detail
import torch.nn.functional as F
p = params
a1 = F.relu(F.batch_norm(taps['conv1'], None, None, p['bn1.weight'], p['bn1.bias'], True, 0.1, 1e-5))
print('a1', rel(S['blocks'][0][1], a1))
blk, xin, stA, h, stB = S['blocks'][0]
oA = F.relu(F.batch_norm(a1, None, None, p['layer1.0.bn1.weight'], p['layer1.0.bn1.bias'], True, 0.1, 1e-5))
h_ref = F.conv2d(oA, p['layer1.0.conv1.weight'], None, 1, 1)
print('h', rel(h, h_ref))
sc_ref = F.conv2d(oA, p['layer1.0.shortcut.0.weight'], None, 1)
oB = F.relu(F.batch_norm(h_ref, None, None, p['layer1.0.bn2.weight'], p['layer1.0.bn2.bias'], True, 0.1, 1e-5))
out_ref = F.conv2d(oB, p['layer1.0.conv2.weight'], None, 1, 1) + sc_ref
print('out0', rel(S['blocks'][1][1], out_ref))
