#!/usr/bin/env python3
"""Golden fixtures synth_eer2_{resnet,ecapa}.npz and init_checksums.npz from the REAL reference.

EER parity in a regime where the reference separates the classes: the reference's
LFCC + ResNet-18 / ECAPA-TDNN-512 + AngularIsoLoss, torch.optim Adam + SGD as configured in
main_train.py:175-176,272 with the reference's own step decay (main_train.py:144-147, --interval 4:
lr = 5e-4 * 0.5^(epoch // 4)), started from the SEEDED construction (torch.manual_seed(688), the modules'
own kaiming initialisers), trained for 16 epochs on a separable synthetic corpus
(asvspoof2021_air_amd/synth.py, mix_lo = 0.4: every spoofed utterance carries at least 40 % of the
artefact), 768 training / 512 held-out 1 s utterances, batch 32.  The reference ends at EER < 1 %.

init_checksums.npz: per-tensor (sum, sum |.|, first element) of the seeded reference state_dicts, so the
drop-in modules' seeded construction can be checked against the reference on the GPU box.

Build container only (needs /root/reference).  Usage: python tests/golden/make_golden_eer2.py [resnet|ecapa|init]
"""
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from make_golden import install_shims, save

L9, B9, NTR, NHO, EPOCHS, INTERVAL, MIX_LO = 16000, 32, 768, 512, 16, 4, 0.4
SEED = 688


def checksums(sd):
    """per tensor: (sum, sum |.|, first element, numel) as float64 and the int64 sum of the raw 32-bit
    patterns (integer addition is associative: a bit-exact, order-independent checksum)."""
    names, vals, bits = [], [], []
    for k, v in sd.items():
        raw = v.detach().reshape(-1)
        v = raw.double()
        names.append(k)
        vals.append([float(v.sum()), float(v.abs().sum()), float(v[0]) if v.numel() else 0.0, float(v.numel())])
        bits.append(int(raw.contiguous().view(torch.int32).long().sum()) if raw.dtype == torch.float32 else int(raw.long().sum()))
    return np.array(names), np.array(vals, dtype=np.float64), np.array(bits, dtype=np.int64)


def build(which):
    import ecapa_tdnn as ref_ecapa  # noqa: E402
    import loss as ref_loss  # noqa: E402
    import resnet as ref_resnet  # noqa: E402
    torch.manual_seed(SEED)
    if which == "resnet":
        net = ref_resnet.ResNet(3, 256, resnet_type="18", nclasses=2)
    else:
        net = ref_ecapa.Res2Net2(ref_ecapa.Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    lossmod = ref_loss.AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    return net, lossmod


def run(which):
    import eval_metrics as ref_em  # noqa: E402
    import feature_extraction as ref_fe  # noqa: E402
    from asvspoof2021_air_amd.synth import corpus
    t0 = time.time()
    pcm_tr, lab_tr = corpus(688, NTR, L9, mix_lo=MIX_LO)
    pcm_ho, lab_ho = corpus(689, NHO, L9, mix_lo=MIX_LO)
    lf = ref_fe.LFCC(320, 160, 512, 16000, 20, with_energy=False)

    def feats_of(pcm):
        with torch.no_grad():
            f = lf(torch.from_numpy(pcm.copy()))  # (n, 101, 60)
        if which == "resnet":
            return f.unsqueeze(1).transpose(2, 3).contiguous()  # main_train.py:338
        return f.transpose(1, 2).contiguous()                  # + squeeze for ECAPA (main_train.py:347)

    xtr, xho = feats_of(pcm_tr), feats_of(pcm_ho)
    net, lossmod = build(which)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0005)
    opt2 = torch.optim.SGD(lossmod.parameters(), lr=5e-4)
    ltr = torch.from_numpy(lab_tr)
    epoch_loss, epoch_eer, step = [], [], 0

    def heldout():
        net.eval()
        sc = []
        with torch.no_grad():
            for i in range(0, NHO, B9):
                torch.manual_seed(9500 + i // B9)  # host-side attention noise (resnet.py:38-42)
                feat, _ = net(xho[i:i + B9])
                _, neg = lossmod(feat, torch.zeros(B9, dtype=torch.long))
                sc.append(-neg)  # generate_score.py:116 writes +cos similarity
        s = torch.cat(sc).numpy()
        e = min(ref_em.compute_eer(s[lab_ho == 0], s[lab_ho == 1])[0],
                ref_em.compute_eer(-s[lab_ho == 0], -s[lab_ho == 1])[0])
        return s, e

    for ep in range(EPOCHS):
        lr = 5e-4 * (0.5 ** (ep // INTERVAL))  # adjust_learning_rate, main_train.py:144-147 / 294-298
        for o in (opt, opt2):
            for gr in o.param_groups:
                gr["lr"] = lr
        net.train()
        tot = 0.0
        for i in range(0, NTR, B9):
            torch.manual_seed(9000 + step)
            feat, _ = net(xtr[i:i + B9])
            loss, _ = lossmod(feat, ltr[i:i + B9])
            opt.zero_grad()
            opt2.zero_grad()
            loss.backward()
            opt.step()
            opt2.step()
            tot += loss.item()
            step += 1
        epoch_loss.append(tot / (NTR // B9))
        scores, eer = heldout()
        epoch_eer.append(eer)
        print("  %s epoch %d loss %.5f held-out EER %.4f (%.0f s)" % (which, ep, epoch_loss[-1], eer, time.time() - t0), flush=True)
    save("synth_eer2_%s.npz" % which, epoch_loss=np.array(epoch_loss), epoch_eer=np.array(epoch_eer), scores=scores,
         labels=lab_ho, eer=np.array(eer), cfg=np.array([L9, B9, NTR, NHO, EPOCHS, INTERVAL]), mix_lo=np.array(MIX_LO),
         seed=np.array(SEED), pcm_sum=np.array([pcm_tr.astype(np.float64).sum(), pcm_ho.astype(np.float64).sum()]))


def init():
    out = {}
    for which in ("resnet", "ecapa"):
        net, lossmod = build(which)
        n, v, bits = checksums(net.state_dict())
        out[which + "_names"], out[which + "_vals"], out[which + "_bits"] = n, v, bits
        out[which + "_center"] = lossmod.center.detach().numpy().copy()
    save("init_checksums.npz", seed=np.array(SEED), **out)


if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(4)
    what = sys.argv[1:] or ["init", "resnet", "ecapa"]
    for w in what:
        init() if w == "init" else run(w)
