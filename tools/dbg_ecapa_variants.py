"""err / oracle-fp32-band per gradient tensor for the ECAPA constructor variants (B = 8, T = 64)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ecapa as o_ecapa, train as o_train
from oracle.filler import fill_module_, fill_state, fill_value, synth_feat
from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
from asvspoof2021_air_amd.loss import AngularIsoLoss
torch.set_num_threads(16)
B, T = 8, 64
xx = synth_feat((B, 60, T), seed=400 + T); labels = (torch.arange(B) % 3 != 0).long()
for ctx, summed in ((True, False), (True, True), (False, True)):
    m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60, context=ctx, summed=summed)
    fill_module_(m); m = m.cuda().train()
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0); fill_module_(lossm); lossm = lossm.cuda()
    feat, _ = m(xx.cuda()); loss, _ = lossm(feat, labels.cuda()); loss.backward()
    sh = o_ecapa.ecapa_shapes(context=ctx)
    p32 = fill_state(sh); p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in p32.items()}
    g64 = o_train.OracleTrainer("ecapa", p64, fill_value("center", (1, 256)).double(), context=ctx, summed=summed).loss_and_grads(xx.double(), labels)[3]
    g32 = o_train.OracleTrainer("ecapa", p32, fill_value("center", (1, 256)), context=ctx, summed=summed).loss_and_grads(xx, labels)[3]
    rows = []
    for k, p in m.named_parameters():
        if g64[k] is None or k in ("attention.2.bias", "attention.3.bias"): continue
        ref = g64[k].numpy(); got = p.grad.cpu().double().numpy(); o32 = g32[k].double().numpy()
        n = np.linalg.norm(ref) + 1e-30
        rows.append((k, np.linalg.norm(got - ref) / n, np.linalg.norm(o32 - ref) / n, np.linalg.norm(got - o32) / n))
    rows.sort(key=lambda r: -r[1])
    print("context=%s summed=%s: HIP-vs-fp64 worst %.3g median %.3g | oracle32-vs-fp64 worst %.3g median %.3g" % (
        ctx, summed, rows[0][1], np.median([r[1] for r in rows]), max(r[2] for r in rows), np.median([r[2] for r in rows])))
    for r in rows[:6]:
        print("   %-28s hip %.3g  oracle32 %.3g  hip-vs-oracle32 %.3g" % r)
