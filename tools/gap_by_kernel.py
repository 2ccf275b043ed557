#!/usr/bin/env python3
"""Where the timeline of a rocprofv3 kernel trace (rocpd .db) goes, per kernel name, over the second half of the
trace: `alone` = time the kernel runs with nothing else on the GPU, `gap` = idle time directly in front of its start
(nothing running).  Small kernels with a large alone + gap are bubbles of the dependent launch chain.
Usage: gap_by_kernel.py results.db [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
rows = rows[len(rows) // 2:]


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return re.sub(r"^void ", "", n)[:70]


# sweep: events sorted by time; track the set of running kernels
ev = []
for i, (n, s, e) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
alone = [0] * len(rows)
gap = [0] * len(rows)
running = set()
last_t = ev[0][0]
idle_since = None
for t, kind, i in ev:
    if len(running) == 1:
        alone[next(iter(running))] += t - last_t
    if kind == 1:
        if not running and idle_since is not None:
            gap[i] += t - idle_since
        running.add(i)
    else:
        running.discard(i)
        if not running:
            idle_since = t
    last_t = t
agg = {}
for i, (n, s, e) in enumerate(rows):
    a = agg.setdefault(short(n), [0, 0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += alone[i]
    a[3] += gap[i]
span = rows[-1][2] - rows[0][1]
print("span %.2f ms, %d kernels; per step (/%g): total idle %.3f ms" % (span / 1e6, len(rows), steps,
                                                                        sum(gap) / 1e6 / steps))
print("| kernel | calls | dur ms | alone ms | gap before ms |")
print("|---|---|---|---|---|")
for k, (c, d, al, g) in sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3]))[:40]:
    print("| %s | %d | %.3f | %.3f | %.3f |" % (k, c, d / 1e6 / steps, al / 1e6 / steps, g / 1e6 / steps))
