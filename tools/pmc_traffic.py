#!/usr/bin/env python3
"""Turn the PMC pass outputs of tools/profile_round.sh (gpurun_out/<tag>/pmc_*_SIZE.txt) into
profiles/<tag>_pmc_traffic.{json,md}: HBM-side bytes per launch for every kernel.
Usage: tools/pmc_traffic.py <tag>"""
import json
import re
import sys

tag = sys.argv[1]
src = "gpurun_out/%s" % tag


def load(path):
    d = {}
    for ln in open(path):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE|SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE)\s+n=(\d+)\s+avg=([\d.e+-]+)", ln)
        if m:
            d[m.group(1).strip()] = (int(m.group(3)), float(m.group(4)))
    return d


out, rows = {}, []
for model in ("resnet", "ecapa"):
    f = load("%s/pmc_%s_FETCH_SIZE.txt" % (src, model))
    w = load("%s/pmc_%s_WRITE_SIZE.txt" % (src, model))
    for k in sorted(f):
        if k not in w:
            continue
        n, fk = f[k]
        _, wk = w[k]
        # MI355X_MICROARCH.md "HBM": counters are KB; on gfx950 FETCH_SIZE tallies 128-byte requests at
        # 64 B -> doubled; WRITE_SIZE as reported
        traffic = (2 * fk + wk) * 1024
        rows.append((model, k, n, fk, wk, traffic))
        out.setdefault(model, {})[k] = {"launches": n, "fetch_kb": fk, "write_kb": wk,
                                        "traffic_bytes_per_launch": traffic}
# per-family aggregates (all template instances of a kernel, launch-weighted) live under their OWN top-level
# key: inside out[model] they would be summed a second time by anyone adding up the per-kernel rows
fam = {}
for model in out:
    # family (= the profiling id's name, csrc/air_prof.h) -> kernel-name prefix of its template instances / variants
    for base, prefix in (("wino4_conv_kernel", "wino4_conv_kernel"), ("wino_conv_kernel", "wino_conv_kernel"),
                         ("wino_wgrad_kernel", "wino_wgrad_kernel"), ("c1b_gemm_kernel", "c1b_gemm"),
                         ("c1b_fwd_kernel", "c1b_fwd")):
        ks = [k for k in out[model] if k.startswith(prefix)]
        n = sum(out[model][k]["launches"] for k in ks)
        if n:
            fam.setdefault(model, {})[base] = {"launches": n, "traffic_bytes_per_launch": sum(
                out[model][k]["launches"] * out[model][k]["traffic_bytes_per_launch"] for k in ks) / n}
# MFMA-pipe busy fraction per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
busy_rows = []
for model in ("resnet", "ecapa"):
    try:
        mb = load("%s/pmc_%s_SQ_VALU_MFMA_BUSY_CYCLES.txt" % (src, model))
        ga = load("%s/pmc_%s_GRBM_GUI_ACTIVE.txt" % (src, model))
    except OSError:
        continue
    for k in sorted(mb):
        if k in ga and ga[k][1] > 0:
            frac = mb[k][1] / (1024.0 * ga[k][1] / 8.0)
            if frac > 0.01:
                busy_rows.append((model, k, mb[k][0], frac))
                out.setdefault(model, {}).setdefault(k, {})["mfma_busy"] = frac
for model in fam:
    for base, v in fam[model].items():
        out[model + "_family"] = fam[model]
json.dump(out, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1, sort_keys=True)
with open("profiles/%s_pmc_traffic.md" % tag, "w") as fh:
    fh.write("# PMC HBM-side traffic per launch (%s), `tools/profile_round.sh` + `tools/pmc_traffic.py`\n\n"
             "Separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over\n"
             "`python bench.py --model <m> --steps 2 --warmup 1 --no-cpu-baseline --no-roofline` (B=64 ResNet fp32 /\n"
             "B=128 ECAPA bf16, T=750).  Units and corrections as MI355X_MICROARCH.md prescribes: counters are KB;\n"
             "FETCH_SIZE is doubled on gfx950 (it tallies 128-byte requests at 64 B); WRITE_SIZE as reported.\n"
             "traffic = 2*FETCH + WRITE, averaged over the launches of the kernel in the run (all layers).  Check:\n"
             "adam_kernel moves 7 x 49.8 MB = 349 MB algorithmically and reports 349 MB.  `bench.py` copies\n"
             "`traffic_bytes_per_launch` of its dominant kernel into `roofline.traffic`.\n\n"
             "| model | kernel | launches | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB/launch |\n|---|---|---|---|---|---|\n" % tag)
    for model, k, n, fk, wk, t in rows:
        if t > 5e6:
            fh.write("| %s | %s | %d | %.4g | %.4g | %.1f |\n" % (model, k[:60], n, fk, wk, t / 1e6))
    if busy_rows:
        fh.write("\n## MFMA pipe busy (separate `--pmc SQ_VALU_MFMA_BUSY_CYCLES` / `--pmc GRBM_GUI_ACTIVE` passes of the same command)\n\n"
                 "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), launch average.\n\n"
                 "| model | kernel | launches | MFMA busy |\n|---|---|---|---|\n")
        for model, k, n, frac in busy_rows:
            fh.write("| %s | %s | %d | %.2f |\n" % (model, k[:60], n, frac))
print(fam)
