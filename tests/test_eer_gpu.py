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
    # (a) training behaviour: the loss curve tracks the reference's.  Epoch 1 is tight; later
    # epochs drift apart step by step (Adam's sign-SGD noise floor) - stated
    # tolerance 20 % per epoch.
    np.testing.assert_allclose(epoch_loss[0], g["epoch_loss"][0], rtol=2e-2)
    np.testing.assert_allclose(epoch_loss, g["epoch_loss"], rtol=0.20)
    # (b) EER parity proper: score the held-out set with the ORACLE using the weights the HIP
    # path just trained.  Same weights -> same scores (1e-3) -> same EER (to one trial).
    from oracle import lfcc as o_lfcc, resnet as o_resnet, eer as o_eer
    from oracle.loss import ocsoftmax_forward
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    centre = lossm.center.detach().cpu()
    xo = torch.from_numpy(o_lfcc.lfcc_forward(pcm_ho.copy())).unsqueeze(1).transpose(2, 3).contiguous()
    o_scores = []
    with torch.no_grad():
        for i in range(0, NHO, B):
            torch.manual_seed(9500 + i // B)
            noise = 1e-5 * torch.randn(B, TA, 256)
            ft, _ = o_resnet.resnet18_forward(sd, xo[i:i + B], training=False, noise=noise)
            o_scores.append(-ocsoftmax_forward(ft, centre, torch.zeros(B, dtype=torch.long), 0.9, 0.2, 20.0)[1])
    o_scores = torch.cat(o_scores).numpy()
    np.testing.assert_allclose(scores, o_scores, atol=1e-3)
    o_eer_val = o_eer.eer_both_polarities(o_scores, lab_ho)
    print("EER  HIP %.4f | oracle on the same weights %.4f | reference's own training %.4f" % (eer, o_eer_val, float(g["eer"])))
    assert abs(eer - o_eer_val) <= 1.0 / 128
    # (c) the reference trained by itself lands in the same regime (this corpus overlaps by
    # construction and 4 epochs are far from convergence, so its EER swings by +-0.1 from epoch
    # to epoch on the reference itself): stated tolerance 0.2 absolute.
    assert abs(eer - float(g["eer"])) <= 0.2, (eer, float(g["eer"]))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_synthetic_corpus_eer_ecapa(golden, dtype):
    """Same check for ECAPA-TDNN-512 (tests/golden/synth_eer_ecapa.npz: the real reference's Res2Net2 trained on
    the same corpus), in the reference's fp32 arithmetic and in bf16 compute (BASELINE configs[2])."""
    g = golden("synth_eer_ecapa.npz")
    L, B, NTR, NHO, EPOCHS = [int(v) for v in g["cfg"]]
    from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
    from asvspoof2021_air_amd.eval_metrics import compute_eer
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.synth import corpus
    from asvspoof2021_air_amd.train import Trainer
    pcm_tr, lab_tr = corpus(688, NTR, L)
    pcm_ho, lab_ho = corpus(689, NHO, L)
    np.testing.assert_array_equal(lab_ho, g["labels"])
    model = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
    fill_module_(model)
    model.set_compute_dtype(dtype)
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    T = 1 + L // 160
    tr = Trainer(model, loss_module=lossm, feat_len=T, ecapa=True)
    xtr, ltr = torch.from_numpy(pcm_tr).cuda(), torch.from_numpy(lab_tr).cuda()
    epoch_loss = []
    for ep in range(EPOCHS):
        tot = 0.0
        for i in range(0, NTR, B):
            loss, _ = tr.step(xtr[i:i + B], ltr[i:i + B])
            tot += loss.item()
        epoch_loss.append(tot / (NTR // B))
    xho = torch.from_numpy(pcm_ho).cuda()
    scores = torch.cat([tr.score(xho[i:i + B]).cpu() for i in range(0, NHO, B)]).numpy()
    eer = min(compute_eer(scores[lab_ho == 0], scores[lab_ho == 1])[0],
              compute_eer(-scores[lab_ho == 0], -scores[lab_ho == 1])[0])
    print(dtype, "epoch losses", epoch_loss, "\nreference        ", list(g["epoch_loss"]),
          "\nEER %.4f vs reference %.4f" % (eer, float(g["eer"])))
    # (a) the loss curve tracks the reference's while that means something: epoch 1 within 2 % (fp32) / 3 % (bf16),
    # epoch 2 within 10 %.  After that the trajectories are chaotic - the CPU oracle ITSELF ends epoch 4 at 1.47
    # with 8 threads and 2.36 with 3 threads (the reference, 8 threads: 1.63) - so epochs 3-4 only have to stay
    # in that band (+-50 %) and keep falling.
    np.testing.assert_allclose(epoch_loss[0], g["epoch_loss"][0], rtol=2e-2 if dtype == "fp32" else 3e-2)
    np.testing.assert_allclose(epoch_loss[1], g["epoch_loss"][1], rtol=0.10)
    np.testing.assert_allclose(epoch_loss[2:], g["epoch_loss"][2:], rtol=0.5)
    assert epoch_loss[3] < epoch_loss[1] < epoch_loss[0]
    # (b) score parity: the ORACLE scores the held-out set with the weights the HIP path trained
    from oracle import ecapa as o_ecapa, lfcc as o_lfcc, eer as o_eer
    from oracle.loss import ocsoftmax_forward
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    centre = lossm.center.detach().cpu()
    xo = torch.from_numpy(o_lfcc.lfcc_forward(pcm_ho.copy())).transpose(1, 2).contiguous()
    o_scores = []
    with torch.no_grad():
        for i in range(0, NHO, B):
            ft, _ = o_ecapa.ecapa_forward(sd, xo[i:i + B], training=False, bf16=(dtype == "bf16"))
            o_scores.append(-ocsoftmax_forward(ft, centre, torch.zeros(B, dtype=torch.long), 0.9, 0.2, 20.0)[1])
    o_scores = torch.cat(o_scores).numpy()
    np.testing.assert_allclose(scores, o_scores, atol=1e-3 if dtype == "fp32" else 1e-2)
    o_eer_val = o_eer.eer_both_polarities(o_scores, lab_ho)
    assert abs(eer - o_eer_val) <= (1.0 / 128 if dtype == "fp32" else 4.0 / 128)
    # (c) the reference trained by itself lands in the same regime (4 epochs are far from convergence)
    assert abs(eer - float(g["eer"])) <= 0.2, (eer, float(g["eer"]))
