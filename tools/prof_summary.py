#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table:
calls, total ms, average us, share.  Usage: prof_summary.py results.db [out.md]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)  # drop the argument list
        name = re.sub(r"^void ", "", name)
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.1f | %.1f |" % (name[:90], n, ns / 1e6, ns / n / 1e3, 100.0 * ns / total))
    lines.append("| TOTAL | %d | %.3f | | 100 |" % (len(rows), total / 1e6))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            fh.write(out + "\n")


if __name__ == "__main__":
    main()
