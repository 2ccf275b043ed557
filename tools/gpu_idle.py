#!/usr/bin/env python3
"""GPU idle fraction in the steady state of a rocprofv3 kernel trace (rocpd .db): is a bench CPU-launch-bound?"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start,end from kernels order by start").fetchall()
n = len(rows); rs = rows[int(n * 0.5):]
busy = sum(e - s for s, e in rs)
# union of intervals (two streams overlap)
cur_s, cur_e, union = rs[0][0], rs[0][1], 0
for s, e in rs[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rs[-1][1] - rs[0][0]
print("%s: kernels %d, span %.1f ms, covered %.1f ms, idle fraction %.3f" % (sys.argv[2] if len(sys.argv) > 2 else "", len(rs), span / 1e6, union / 1e6, 1 - union / span))
