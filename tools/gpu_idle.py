#!/usr/bin/env python3
"""GPU idle fraction in the steady state of a rocprofv3 kernel trace (rocpd .db): is a bench CPU-launch-bound?
Also prints the kernel time per HIP stream / queue (which stream carries the critical path)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start,end from kernels order by start").fetchall()
n = len(rows); rs = rows[int(n * 0.5):]
busy = sum(e - s for s, e in rs)
# union of intervals (two streams overlap)
cur_s, cur_e, union = rs[0][0], rs[0][1], 0
gaps = []
for s, e in rs[1:]:
    if s > cur_e:
        union += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rs[-1][1] - rs[0][0]
print("%s: kernels %d, span %.1f ms, covered %.1f ms, sum of kernel time %.1f ms, idle fraction %.3f" % (
    sys.argv[2] if len(sys.argv) > 2 else "", len(rs), span / 1e6, union / 1e6, busy / 1e6, 1 - union / span))
gaps.sort()
if gaps:
    print("gaps: %d, median %.1f us, p90 %.1f us, max %.1f us, total %.2f ms" % (
        len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3, sum(gaps) / 1e6))
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
for key in ("stream_id", "queue_id", "stream", "queue"):
    if key in cols:
        t0 = rs[0][0]
        for k, cnt, tot in db.execute("select %s, count(*), sum(end-start) from kernels where start >= ? group by %s" % (key, key), (t0,)):
            print("  %s %s: %d kernels, %.1f ms" % (key, k, cnt, tot / 1e6))
        break
