#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_wino4.txt (tools/pmc_wino4.sh) -> the per-layer table of profiles/<round>_wino4_layers.md.
Usage: tools/wino4_layers_md.py <tag>"""
import re
import sys

txt = open("gpurun_out/%s/pmc_wino4.txt" % sys.argv[1]).read()
blocks = re.split(r"== (l\d) (\w) \[.*?\]\n", txt)
res = {}
it = iter(blocks[1:])
for layer, pas, body in zip(it, it, it):
    d = res.setdefault((layer, pas), {})
    for ln in body.splitlines():
        m = re.match(r"(wino4_conv_kernel.*?)\s+(FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE)\s+n=(\d+)\s+avg=([\d.e+-]+)", ln)
        if m:
            d[m.group(2)] = float(m.group(4))
SH = {"l1": (64, 18, 750, 64), "l2": (128, 9, 375, 128), "l3": (256, 5, 188, 256), "l4": (512, 3, 94, 512)}
print("| layer pass | fetch MB | algorithmic read MB | ratio | write MB | L2 hit | MFMA busy |\n|---|---|---|---|---|---|---|")
for (layer, pas), d in res.items():
    C, H, W, Co = SH[layer]
    B = 64
    x, y, u = B * C * H * W * 4, B * Co * H * W * 4, Co * C * 32 * 4
    alg = (x + u + (y if pas == "r" else 0)) / 1e6
    f, w = 2 * d["FETCH_SIZE"] * 1024 / 1e6, d["WRITE_SIZE"] * 1024 / 1e6
    hit = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
    print("| %s %s | %.0f | %.0f | %.2f | %.0f | %.2f | %.2f |" % (layer, pas, f, alg, f / alg, w, hit, busy))
