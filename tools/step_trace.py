#!/usr/bin/env python
"""Per-launch listing of ONE training step from a rocprofv3 rocpd sqlite database (kernel-trace):
the launches between the last two `adam_tf` kernels, in start order, with grid size, duration and
the idle gap to the previous kernel.  Usage: step_trace.py <db> [out.txt]"""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    tab = "kernels" if "kernels" in tables else [t for t in tables if "kernel" in t][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % tab)]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in ("grid_x", "grid_y", "grid_z", "workgroup_x", "grid_size_x", "grid_size_y", "grid_size_z",
                         "workgroup_size_x", "lds_size", "lds_block_size") if c in cols]
    rows = cur.execute("select %s, start, end%s from %s order by start" %
                       (name_col, "".join(", " + c for c in gcols), tab)).fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_tf" in r[0]]
    lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
    lines = ["columns: idx dur_us gap_us %s name" % " ".join(gcols)]
    prev_end = rows[lo - 1][2] if lo > 0 else rows[lo][1]
    tot = 0
    for i in range(lo, hi):
        r = rows[i]
        dur = (r[2] - r[1]) / 1e3
        gap = (r[1] - prev_end) / 1e3
        prev_end = max(prev_end, r[2])
        tot += dur
        nm = r[0].split("(")[0]
        nm = nm.replace("faststyle::", "").replace("void ", "")
        lines.append("%4d %9.2f %7.2f %s  %s" % (i - lo, dur, gap, " ".join(str(x) for x in r[3:]), nm[:80]))
    lines.append("launches %d, kernel time %.1f us, span %.1f us" % (hi - lo, tot, (rows[hi - 1][2] - rows[lo][1]) / 1e3))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
