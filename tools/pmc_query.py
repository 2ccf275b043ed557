#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from a rocprofv3 rocpd database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
pm = [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()]
view = 'counters_collection' if 'counters_collection' in tabs else None
if view is None:
    print(pm); sys.exit(0)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
rows = cur.execute("select kernel_name, counter_name, value from %s" % view).fetchall() if 'kernel_name' in cols else []
agg = {}
for k, c, v in rows:
    k = re.sub(r"\(.*$", "", k.replace("(anonymous namespace)::", "")).replace("void ", "")
    a = agg.setdefault((k, c), [0, 0.0]); a[0] += 1; a[1] += v
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, c), (n, s) in sorted(agg.items()):
    if filt in k: print("%-50s %-28s n=%-4d avg=%.4g" % (k[:50], c, n, s / n))
