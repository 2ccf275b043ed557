#!/usr/bin/env python3
"""Golden fixture synth_eer_ecapa.npz: the REAL reference (feature_extraction.LFCC, ecapa_tdnn.Res2Net2,
loss.AngularIsoLoss, torch.optim Adam + SGD as configured in main_train.py:175-176,272) trained on the
synthetic corpus of asvspoof2021_air_amd/synth.py, under the shims of make_golden.py.  Per-epoch mean loss,
held-out scores and EER: the "reference EER" the ECAPA drop-in must match (ECAPA variant).
Build container only (needs /root/reference).  Usage: python tests/golden/make_golden_ecapa_eer.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from make_golden import install_shims, save


def main():
    install_shims()
    torch.set_num_threads(8)
    import ecapa_tdnn as ref_ecapa  # noqa: E402
    import eval_metrics as ref_em  # noqa: E402
    import feature_extraction as ref_fe  # noqa: E402
    import loss as ref_loss  # noqa: E402
    from asvspoof2021_air_amd.synth import corpus
    from oracle.filler import fill_module_
    L9, B9, NTR, NHO, EPOCHS = 16000, 32, 384, 256, 4
    pcm_tr, lab_tr = corpus(688, NTR, L9)
    pcm_ho, lab_ho = corpus(689, NHO, L9)
    lf = ref_fe.LFCC(320, 160, 512, 16000, 20, with_energy=False)

    def feats_of(pcm):
        with torch.no_grad():
            f = lf(torch.from_numpy(pcm.copy()))  # (n, 101, 60)
        return f.transpose(1, 2).contiguous()  # (n, 60, 101): main_train.py:338 + :347 (squeeze for ECAPA)

    xtr, xho = feats_of(pcm_tr), feats_of(pcm_ho)
    net = ref_ecapa.Res2Net2(ref_ecapa.Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    fill_module_(net)
    lossmod = ref_loss.AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossmod)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0005)
    opt2 = torch.optim.SGD(lossmod.parameters(), lr=5e-4)
    ltr = torch.from_numpy(lab_tr)
    epoch_loss = []
    for ep in range(EPOCHS):
        net.train()
        tot = 0.0
        for i in range(0, NTR, B9):
            feat, _ = net(xtr[i:i + B9])
            loss, _ = lossmod(feat, ltr[i:i + B9])
            opt.zero_grad()
            opt2.zero_grad()
            loss.backward()
            opt.step()
            opt2.step()
            tot += loss.item()
        epoch_loss.append(tot / (NTR // B9))
        print("  epoch", ep, epoch_loss[-1], flush=True)
    net.eval()
    scores = []
    with torch.no_grad():
        for i in range(0, NHO, B9):
            feat, _ = net(xho[i:i + B9])
            _, neg = lossmod(feat, torch.zeros(B9, dtype=torch.long))
            scores.append(-neg)
    scores = torch.cat(scores).numpy()
    eer = min(ref_em.compute_eer(scores[lab_ho == 0], scores[lab_ho == 1])[0],
              ref_em.compute_eer(-scores[lab_ho == 0], -scores[lab_ho == 1])[0])
    print("  epoch losses", epoch_loss, " held-out EER %.4f" % eer)
    save("synth_eer_ecapa.npz", epoch_loss=np.array(epoch_loss), scores=scores, labels=lab_ho, eer=np.array(eer),
         cfg=np.array([L9, B9, NTR, NHO, EPOCHS]))


if __name__ == "__main__":
    main()
