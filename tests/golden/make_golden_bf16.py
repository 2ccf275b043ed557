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

Round 4 adds the EVAL-mode runs (running statistics: no amplification), which pin the arithmetic sharply.

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
    # ---- eval mode (round 4): running statistics instead of batch statistics - no ~100x
    # amplification of the rounding, so HERE a sharp bound holds between any two faithful bf16 evaluations of the
    # graph.  The real reference module under autocast, eval(), same filler weights and buffers; next to it the fp32
    # reference and both oracle modes.  tests/test_oracle_golden.py and tests/test_ecapa_gpu.py assert the oracle
    # and the HIP eval forward against ``feat_autocast_eval_*`` (not against the fp32 golden).
    for tag, (B, T) in (("small", (2, 96)), ("full", (8, 750))):
        x = synth_feat((B, 60, T), seed=900 + T)
        net = ref_ecapa.Res2Net2(ref_ecapa.Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        fill_module_(net)
        net.train(False)
        caps = {}
        hooks = [getattr(net, n).register_forward_hook(lambda m, i, o, n=n: caps.__setitem__(n, o.detach()))
                 for n in ("bn1", "layer1")]
        with torch.no_grad():
            f32e, o32e = net(x)
            c32 = dict(caps)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                face, oace = net(x)
        for h in hooks:
            h.remove()
        assert face.dtype == torch.bfloat16 and caps["bn1"].dtype == torch.bfloat16
        out["feat_autocast_eval_" + tag] = face.float()
        out["out_autocast_eval_" + tag] = oace.float()
        out["feat_fp32_eval_" + tag] = f32e
        if tag == "small":
            # Where the two rule sets COINCIDE a sharp pin exists.  First layer (conv1 -> ReLU -> bn1, :159-161): both
            # round the operands to bf16, accumulate in fp32, round the conv output, and round the BatchNorm output -
            # so utterance 0's (512, 96) bf16 tensor must reproduce bit for bit up to fp32 summation order (a handful
            # of values on a rounding boundary).  Stored as bf16 BITS.  Behind the first Bottle2neck the rule sets
            # differ (SE product and residual sum: two roundings under autocast, one here; squeeze mean in bf16 vs
            # fp32): every 4th channel of utterance 0 as the sample for a relative-L2 bound.
            out["h0_bits_autocast_eval_small"] = caps["bn1"][0].contiguous().view(torch.int16).numpy().view(np.uint16)
            out["x1_sample_autocast_eval_small"] = caps["layer1"][0, ::4].float()
            out["x1_sample_fp32_eval_small"] = c32["layer1"][0, ::4].float()
        eparams = {k: v.detach().clone() for k, v in net.state_dict().items()}
        msg = []
        for mode in modes:
            taps = {}
            fo, _ = o_ecapa.ecapa_forward(eparams, x, training=False, bf16=mode, taps=taps)
            msg.append("oracle(bf16=%s) vs autocast %.3g, vs fp32 %.3g" % (mode, rel_l2(fo, face.float()), rel_l2(fo, f32e)))
            if "h0" in taps and tag == "small":
                def mismatches(h0):
                    hb = h0[0].to(torch.bfloat16)
                    assert torch.equal(hb.float(), h0[0].float()), "resident h0 is not bf16-representable"
                    return int((hb.view(torch.int16) != caps["bn1"][0].view(torch.int16)).sum()), hb.numel()
                n_prod, n_all = mismatches(taps["h0"])
                o_ecapa.AUTOCAST_BIAS = True   # autocast's rule for the conv bias (oracle/ecapa.py)
                try:
                    t2 = {}
                    o_ecapa.ecapa_forward(eparams, x, training=False, bf16=mode, taps=t2)
                finally:
                    o_ecapa.AUTOCAST_BIAS = False
                n_auto, _ = mismatches(t2["h0"])
                msg.append("h0 (first layer's stored output): %d of %d values differ from autocast with the bias rounded as "
                           "autocast does, %d with this build's fp32 bias; x1 vs autocast %.3g (autocast vs fp32 %.3g)" % (
                               n_auto, n_all, n_prod, rel_l2(taps["x1"][0, ::4], caps["layer1"][0, ::4].float()),
                               rel_l2(caps["layer1"][0, ::4].float(), c32["layer1"][0, ::4])))
        print("  eval %s: autocast vs fp32 %.3g; %s" % (tag, rel_l2(face.float(), f32e), "; ".join(msg)))
    save("ecapa_bf16.npz", **out)


if __name__ == "__main__":
    main()
