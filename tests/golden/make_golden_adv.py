#!/usr/bin/env python3
"""Golden fixture adv.npz for the adversarial channel-classifier branch, from the
REAL reference ``model.ChannelClassifier`` / ``GradientReversal`` (model.py:976-1023) and
``nn.CrossEntropyLoss`` (main_train.py:251), imported under the shims of make_golden.py.
Build container only (needs /root/reference).  Usage: python tests/golden/make_golden_adv.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np
import torch

from make_golden import install_shims, maxabs, save


def main():
    install_shims()
    import model as ref_model  # noqa: E402
    from oracle import adversarial as o_adv
    from oracle.filler import fill_module_, fill_state, synth_feat
    B, ENC, NC, LAM = 16, 256, 10, 0.05
    clf = ref_model.ChannelClassifier(ENC, NC, LAM)
    fill_module_(clf)
    assert {k: tuple(v.shape) for k, v in clf.state_dict().items()} == o_adv.classifier_shapes(ENC, NC)
    feats = synth_feat((B, ENC), seed=901)
    labels = torch.from_numpy(np.random.Generator(np.random.PCG64(902)).integers(0, NC, B))
    crit = torch.nn.CrossEntropyLoss()
    out = {"feats": feats, "labels": labels, "cfg": np.array([B, ENC, NC]), "lambda": np.array(LAM)}
    params = fill_state(o_adv.classifier_shapes(ENC, NC))
    for mode in ("eval", "train"):
        clf.train(mode == "train")
        clf.zero_grad()
        keep = None
        if mode == "train":
            torch.manual_seed(903)
            keep = torch.nn.functional.dropout(torch.ones(B, ENC // 2), 0.3, True)  # the mask nn.Dropout will draw
            torch.manual_seed(903)
        f = feats.clone().requires_grad_(True)
        logits = clf(f)
        loss = crit(logits, labels)
        loss.backward()
        lo, lg, df, gr = o_adv.loss_and_grads(params, feats, labels, LAM, keep)
        print(mode, "oracle vs reference: loss %.2e logits %.2e dfeats %.2e dW1 %.2e dW2 %.2e" % (
            abs(lo.item() - loss.item()), maxabs(lg, logits), maxabs(df, f.grad),
            maxabs(gr["classifier.0.weight"], clf.classifier[0].weight.grad),
            maxabs(gr["classifier.3.weight"], clf.classifier[3].weight.grad)))
        assert maxabs(lg, logits) < 1e-6 and maxabs(df, f.grad) < 1e-8
        out.update({"logits_" + mode: logits, "loss_" + mode: loss.detach(), "dfeats_" + mode: f.grad,
                    "dw1_" + mode: clf.classifier[0].weight.grad, "db1_" + mode: clf.classifier[0].bias.grad,
                    "dw2_" + mode: clf.classifier[3].weight.grad, "db2_" + mode: clf.classifier[3].bias.grad})
        if keep is not None:
            out["keep"] = keep
    save("adv.npz", **out)


if __name__ == "__main__":
    main()
