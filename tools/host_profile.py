"""Where the HOST time of a train step goes (cProfile over eager steps; the GPU runs behind)."""
import cProfile, pstats, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from asvspoof2021_air_amd.train import Trainer
which = sys.argv[1] if len(sys.argv) > 1 else "ecapa"
dev = torch.device("cuda", 0)
torch.manual_seed(688)
if which == "ecapa":
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    model = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60); model.set_compute_dtype("bf16"); bench.BATCH = 128
else:
    from asvspoof2021_air_amd.resnet import ResNet
    model = ResNet(3, 256, resnet_type="18", nclasses=2); bench.BATCH = 64
tr = Trainer(model, enc_dim=256, lr=5e-4, r_real=0.9, r_fake=0.2, alpha=20.0, feat_len=750, device=dev, ecapa=(which == "ecapa"))
batches = [bench.synth_batch(i, 0, dev) for i in range(2)]
for i in range(5):
    tr.step(*batches[i % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    tr.step(*batches[i % 2])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
