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
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+)\s+avg=([\d.e+-]+)", ln)
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
ks = [k for k in out["resnet"] if k.startswith("wino_conv_kernel")]
n = sum(out["resnet"][k]["launches"] for k in ks)
out["resnet"]["wino_conv_kernel"] = {"launches": n, "traffic_bytes_per_launch": sum(
    out["resnet"][k]["launches"] * out["resnet"][k]["traffic_bytes_per_launch"] for k in ks) / n}
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
print(out["resnet"]["wino_conv_kernel"], out["ecapa"].get("c1b_gemm_kernel"))
