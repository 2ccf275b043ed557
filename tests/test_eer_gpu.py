"""GPU: held-out EER on the synthetic corpus vs the REFERENCE trained on the same data
(tests/golden/synth_eer.npz, produced by the real reference modules in make_golden.py G9):
PCM -> fused HIP LFCC -> ResNet-18 -> OC-Softmax, Adam + SGD, 4 epochs x 12 steps, batch 32."""
import numpy as np
import pytest
import torch

from oracle.filler import fill_module_

pytestmark = pytest.mark.gpu


def test_synthetic_corpus_eer_matches_reference(golden):
    g = golden("synth_eer.npz")
    L, B, NTR, NHO, EPOCHS = [int(v) for v in g["cfg"]]
    from asvspoof2021_air_amd.eval_metrics import compute_eer
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.resnet import ResNet
    from asvspoof2021_air_amd.synth import corpus
    from asvspoof2021_air_amd.train import Trainer
    pcm_tr, lab_tr = corpus(688, NTR, L)
    pcm_ho, lab_ho = corpus(689, NHO, L)
    np.testing.assert_allclose([pcm_tr.astype(np.float64).sum(), pcm_ho.astype(np.float64).sum()], g["pcm_sum"], rtol=1e-9)
    np.testing.assert_array_equal(lab_ho, g["labels"])
    model = ResNet(3, 256, resnet_type="18", nclasses=2)
    fill_module_(model)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    T = 1 + L // 160
    tr = Trainer(model, loss_module=lossm, feat_len=T)
    xtr = torch.from_numpy(pcm_tr).cuda()
    ltr = torch.from_numpy(lab_tr).cuda()
    TA = T
    for _ in range(3):
        TA = (TA + 2 - 3) // 2 + 1
    epoch_loss, step = [], 0
    for ep in range(EPOCHS):
        tot = 0.0
        for i in range(0, NTR, B):
            torch.manual_seed(9000 + step)  # replay the reference's host-side attention noise
            model.set_attention_noise(1e-5 * torch.randn(B, TA, 256))
            loss, _ = tr.step(xtr[i:i + B], ltr[i:i + B])
            tot += loss.item()
            step += 1
        epoch_loss.append(tot / (NTR // B))
    scores = []
    xho = torch.from_numpy(pcm_ho).cuda()
    for i in range(0, NHO, B):
        torch.manual_seed(9500 + i // B)
        model.set_attention_noise(1e-5 * torch.randn(B, TA, 256))
        scores.append(tr.score(xho[i:i + B]).cpu())
    scores = torch.cat(scores).numpy()
    eer = min(compute_eer(scores[lab_ho == 0], scores[lab_ho == 1])[0],
              compute_eer(-scores[lab_ho == 0], -scores[lab_ho == 1])[0])
    print("epoch losses", epoch_loss, "\nreference   ", list(g["epoch_loss"]), "\nEER %.4f vs reference %.4f" % (eer, float(g["eer"])))
    # first epoch (before trajectories can diverge) is tight; afterwards Adam's sign-SGD noise
    # floor makes two fp32 implementations drift apart step by step
    np.testing.assert_allclose(epoch_loss[0], g["epoch_loss"][0], rtol=2e-2)
    np.testing.assert_allclose(epoch_loss, g["epoch_loss"], rtol=0.15)  # stated tolerance: 15 % per epoch
    assert abs(eer - float(g["eer"])) <= 0.05, (eer, float(g["eer"]))  # stated tolerance: 5 points
    corr = np.corrcoef(scores, g["scores"])[0, 1]
    print("score correlation with the reference's held-out scores: %.3f" % corr)
    assert corr > 0.7
