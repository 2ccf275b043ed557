#!/usr/bin/env python3
"""Golden fixtures for the NON-DEFAULT constructor options of the reference's ``Res2Net2`` (ecapa_tdnn.py:99):
``context=False`` (attention sees layer4's output only, :126-129 / :177-180) and ``summed=True`` (layer2 / layer3 read
x + x1 / x + x1 + x2, :163-166) - the variants the reference's own score files were made with
(lfcc_ecapa512c{t,f}s{t,f}_*).  Same recipe as make_golden.py's G5: the REAL reference (imported read-only under the
shims of make_golden.py), filler weights, seeded input; train-mode feat / out, the OC-Softmax loss and every parameter's
gradient norm + four whole gradient tensors.  The oracle (oracle/ecapa.py, ``context=`` / ``summed=``) is checked against
the reference while the fixture is written.

Usage:  python tests/golden/make_golden_ecapa_variants.py   (build container only: needs /root/reference)"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import torch

from make_golden import install_shims, maxabs, save


def main():
    install_shims()
    import ecapa_tdnn as ref_ecapa  # noqa: E402
    import loss as ref_loss  # noqa: E402
    from oracle import ecapa as o_ecapa, train as o_train
    from oracle.filler import fill_module_, fill_value, synth_feat
    torch.set_num_threads(8)
    out = {}
    B, T = 2, 96
    x = synth_feat((B, 60, T), seed=400 + T)
    labels = torch.tensor([0, 1])
    for ctx, summed, enc in ((False, False, "ECA"), (True, True, "ECA"), (False, True, "ECA"), (True, False, "ASP"),
                             (False, True, "ASP")):
        tag = "c%ss%s" % ("t" if ctx else "f", "t" if summed else "f") + ("" if enc == "ECA" else "_asp")
        net = ref_ecapa.Res2Net2(ref_ecapa.Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60, context=ctx, summed=summed,
                                 encoder_type=enc)
        fill_module_(net)
        ref_shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        assert ref_shapes == [(k, tuple(v)) for k, v in o_ecapa.ecapa_shapes(context=ctx, encoder_type=enc).items()], tag
        params = {k: v.detach().clone() for k, v in net.state_dict().items()}
        for mode in ("train", "eval"):
            net.train(mode == "train")
            fill_module_(net)
            feat, o = net(x)
            fo, oo = o_ecapa.ecapa_forward(params, x, training=(mode == "train"), context=ctx, summed=summed)
            print("%s/%s: feat oracle-vs-ref %.3g out %.3g |feat|max %.3g" % (tag, mode, maxabs(fo, feat), maxabs(oo, o),
                                                                          feat.abs().max().item()))
            assert maxabs(fo, feat) <= 2e-5 and maxabs(oo, o) <= 2e-4
            out["feat_%s_%s" % (tag, mode)] = feat.detach()
            out["out_%s_%s" % (tag, mode)] = o.detach()
        net.train(True)
        fill_module_(net)
        feat, _ = net(x)
        lossmod = ref_loss.AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
        fill_module_(lossmod)
        loss, _ = lossmod(feat, labels)
        loss.backward()
        tr = o_train.OracleTrainer("ecapa", params, fill_value("center", (1, 256)), context=ctx, summed=summed)
        lo, _, _, go, gco, _ = tr.loss_and_grads(x, labels)
        gerr = 0.0
        for k, pp in net.named_parameters():
            if pp.grad is None:
                assert go[k] is None, k
                continue
            gerr = max(gerr, maxabs(go[k], pp.grad) / (pp.grad.abs().max().item() + 1e-12))
            out["gnorm_%s_%s" % (tag, k)] = pp.grad.norm()
        print("%s grads: loss %.6f oracle-vs-ref %.3g, worst rel grad err %.3g, centre %.3g" % (
            tag, loss.item(), abs(lo.item() - loss.item()), gerr, maxabs(gco, lossmod.center.grad)))
        assert abs(lo.item() - loss.item()) <= 1e-5 * abs(loss.item()) and gerr <= 5e-2  # (B = 2: stiff, see test_ecapa_gpu)
        out["loss_" + tag] = loss.detach()
        out["g_%s_conv1.bias" % tag] = net.conv1.bias.grad
        out["g_%s_layer2.conv1.weight" % tag] = net.layer2.conv1.weight.grad[:8].clone()
        out["g_%s_attention.0.weight" % tag] = net.attention[0].weight.grad[:4].clone()
        out["g_%s_center" % tag] = lossmod.center.grad
    save("ecapa_variants.npz", **out)


if __name__ == "__main__":
    main()
