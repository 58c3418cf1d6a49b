#!/usr/bin/env python
"""Per-launch listing of ONE stylized frame from a rocprofv3 --kernel-trace --output-format csv directory: the launches between the last two
`apply_tanh` kernels (the last kernel of a forward pass; FS_TRACE_MARK=<substring> names another end-of-unit kernel, e.g. adam_tf for a train step), in start order, with grid size, duration and the idle gap to the previous kernel; every
column is the median over the last FRAMES frames.  Usage: frame_trace.py <dir> [out.txt] [frames]"""
import csv
import glob
import os
import statistics
import sys


def main(d, out=None, frames=8):
    path = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                     int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort(key=lambda r: r[1])
    mark = os.environ.get("FS_TRACE_MARK", "apply_tanh")
    ends = [i for i, r in enumerate(rows) if mark in r[0]]
    spans = [(ends[k] + 1, ends[k + 1] + 1) for k in range(len(ends) - 1)][-frames:]
    n = spans[-1][1] - spans[-1][0]
    spans = [s for s in spans if s[1] - s[0] == n]
    lines = ["columns: idx dur_us gap_us workgroups name   (median of %d frames)" % len(spans)]
    tot = gaps = 0.0
    for j in range(n):
        durs, gp = [], []
        for lo, hi in spans:
            r = rows[lo + j]
            durs.append((r[2] - r[1]) / 1e3)
            gp.append((r[1] - rows[lo + j - 1][2]) / 1e3)
        r = rows[spans[-1][0] + j]
        nm = r[0].split("(")[0].replace("void ", "").replace("fs::", "")
        du, ga = statistics.median(durs), statistics.median(gp)
        tot += du
        gaps += max(ga, 0.0)
        lines.append("%4d %9.2f %7.2f %6d  %s" % (j, du, ga, r[3] // max(r[4], 1), nm[:90]))
    wall = statistics.median([(rows[hi - 1][2] - rows[lo - 1][2]) / 1e3 for lo, hi in spans])
    lines.append("launches %d, kernel time %.1f us, gaps %.1f us, frame (end of the previous frame's last kernel -> end of this one's) %.1f us" % (n, tot, gaps, wall))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 8)
