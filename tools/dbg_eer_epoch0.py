import sys, numpy as np, torch
sys.path.insert(0, "tests")
from asvspoof2021_air_amd.synth import corpus
from asvspoof2021_air_amd.resnet import ResNet
from asvspoof2021_air_amd.loss import AngularIsoLoss
from asvspoof2021_air_amd.train import Trainer
g = np.load("tests/golden/synth_eer2_resnet.npz")
L, B = 16000, 32
pcm, lab = corpus(688, 96, L, mix_lo=0.4)
torch.manual_seed(688)
model = ResNet(3, 256, resnet_type="18", nclasses=2); lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
tr = Trainer(model, loss_module=lossm, feat_len=101)
x, l = torch.from_numpy(pcm).cuda(), torch.from_numpy(lab).cuda()
out = []
for step in range(3):
    torch.manual_seed(9000 + step); model.set_attention_noise(1e-5 * torch.randn(B, 13, 256))
    loss, _ = tr.step(x[step*B:(step+1)*B], l[step*B:(step+1)*B]); out.append(loss.item())
print("first 3 step losses", out)
