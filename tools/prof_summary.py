#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(same columns as `rocprofv3 --stats`: calls, total, average, min, max, percentage)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["%-90s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalDur(ns)", "AvgDur(ns)", "MinDur(ns)", "MaxDur(ns)", "Pct")]
    for n, c, t, a, mn, mx in rows:
        lines.append("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:90], c, t, a, mn, mx, 100.0 * t / tot))
    lines.append("TOTAL kernel time (ns): %d" % tot)
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
