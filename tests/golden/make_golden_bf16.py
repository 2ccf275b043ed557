#!/usr/bin/env python3
"""Pin for BASELINE configs[2] (ECAPA-TDNN-512 "bf16 train"): the REAL reference ``ecapa_tdnn.Res2Net2`` run under
``torch.autocast('cpu', dtype=torch.bfloat16)`` on the closed-form filler weights, next to the same module in fp32,
next to this repo's bf16 arithmetic (``oracle/ecapa.py``, ``bf16=...``).

The reference itself has no reduced-precision path, so "what bf16 training of it computes" is
autocast's rule set applied to its graph.  This script records, for (B, T) = (2, 96) and (8, 750):

* ``feat`` / ``out`` / the OC-Softmax loss of the autocast run (the loss module runs outside autocast on
  ``feat.float()``, as a trainer that keeps its head in fp32 would) and the per-tensor gradient norms;
* the same from the fp32 reference - the scale of "how far bf16 moves anything";
* relative L2 distances oracle-vs-autocast, per tensor, for every oracle bf16 mode.

CPU autocast rounds to bf16 at: conv1d (Conv1d weight AND activations, output tensor bf16), linear (fc6, fc7);
``batch_norm``, ``relu``, ``sigmoid``, ``mean`` / ``var`` / ``sqrt`` / ``clamp``, ``cat`` / ``split`` / add / mul and
``softmax`` / ``sum`` (CPU autocast has no fp32 list entry for them) follow their input dtype, i.e. every (B, C, T)
activation between layers IS a bf16 tensor and every elementwise result is rounded to bf16.  Differences of this
repo's arithmetic from that rule set are listed in DESIGN.md section 2 from the output of this script.

Usage:  python tests/golden/make_golden_bf16.py        (build container only: needs /root/reference)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

from make_golden import install_shims, save  # noqa: E402


def rel_l2(a, b):
    a = a.detach().double().numpy().ravel()
    b = b.detach().double().numpy().ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def main():
    install_shims()
    import ecapa_tdnn as ref_ecapa  # noqa: E402
    import loss as ref_loss  # noqa: E402
    from oracle import ecapa as o_ecapa, train as o_train
    from oracle.filler import fill_module_, fill_value, synth_feat

    torch.set_num_threads(8)
    out = {}
    modes = [m for m in getattr(o_ecapa, "BF16_MODES", (True,))]
    for tag, (B, T) in (("small", (2, 96)), ("full", (8, 750))):
        x = synth_feat((B, 60, T), seed=900 + T)
        labels = (torch.arange(B) % 3 != 0).long()
        runs = {}
        for kind in ("fp32", "autocast"):
            net = ref_ecapa.Res2Net2(ref_ecapa.Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
            fill_module_(net)
            net.train(True)
            lossmod = ref_loss.AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
            fill_module_(lossmod)
            if kind == "autocast":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    feat, o = net(x)
                assert feat.dtype == torch.bfloat16, feat.dtype
            else:
                feat, o = net(x)
            loss, _ = lossmod(feat.float(), labels)
            loss.backward()
            grads = {k: p.grad.detach().float().clone() for k, p in net.named_parameters() if p.grad is not None}
            runs[kind] = (feat.detach().float(), o.detach().float(), loss.detach(), grads)
            print("  %s/%s: loss %.6f |feat|max %.3f" % (tag, kind, loss.item(), feat.float().abs().max().item()))
        f32, fac = runs["fp32"], runs["autocast"]
        out["x_seed_" + tag] = np.array([900 + T, B, T])
        out["feat_autocast_" + tag] = fac[0]
        out["out_autocast_" + tag] = fac[1]
        out["loss_autocast_" + tag] = fac[2]
        out["loss_fp32_" + tag] = f32[2]
        out["feat_fp32_" + tag] = f32[0]
        names = sorted(fac[3])
        out["grad_names_" + tag] = np.array(names)
        out["gnorm_autocast_" + tag] = np.array([fac[3][k].norm().item() for k in names])
        out["gnorm_fp32_" + tag] = np.array([f32[3][k].norm().item() for k in names])
        d = np.array([rel_l2(fac[3][k], f32[3][k]) for k in names])
        out["grad_rel_autocast_vs_fp32_" + tag] = d
        print("  %s: autocast vs fp32: feat rel-L2 %.3g, loss %.3g, grads median %.3g max %.3g" % (
            tag, rel_l2(fac[0], f32[0]), abs(fac[2].item() / f32[2].item() - 1.0), np.median(d), d.max()))
        # this repo's bf16 arithmetic, every mode the oracle offers, against both reference runs
        net = ref_ecapa.Res2Net2(ref_ecapa.Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        fill_module_(net)
        eparams = {k: v.detach().clone() for k, v in net.state_dict().items()}
        for mode in modes:
            tr = o_train.OracleTrainer("ecapa", eparams, fill_value("center", (1, 256)), bf16=mode)
            lo, _, fo, go, _, _ = tr.loss_and_grads(x, labels)
            da = np.array([rel_l2(go[k], fac[3][k]) for k in names])
            df = np.array([rel_l2(go[k], f32[3][k]) for k in names])
            mtag = "%s_%s" % (str(mode).lower(), tag)
            out["oracle_feat_rel_autocast_" + mtag] = np.array(rel_l2(fo, fac[0]))
            out["oracle_feat_rel_fp32_" + mtag] = np.array(rel_l2(fo, f32[0]))
            out["oracle_loss_" + mtag] = lo
            out["oracle_grad_rel_autocast_" + mtag] = da
            out["oracle_grad_rel_fp32_" + mtag] = df
            print("  %s oracle(bf16=%s): feat vs autocast %.3g (vs fp32 %.3g), loss %.6f, grads vs autocast median %.3g "
                  "max %.3g (vs fp32 median %.3g max %.3g)" % (tag, mode, rel_l2(fo, fac[0]), rel_l2(fo, f32[0]), lo.item(),
                                                              np.median(da), da.max(), np.median(df), df.max()))
    save("ecapa_bf16.npz", **out)


if __name__ == "__main__":
    main()
