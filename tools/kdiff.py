#!/usr/bin/env python
"""Per-step comparison of two kernel_stats tables (tools/kstats.sh / tools/collect_profiles.sh): microseconds and launches per train step of every
kernel symbol, steps = launches of fs::wgw_kernel (one per step).   python tools/kdiff.py <old.txt> <new.txt> [min_us]"""
import re
import sys


def load(p):
    rows = {}
    for line in open(p).read().splitlines()[1:]:
        m = re.match(r"(.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)%$", line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), int(m.group(3)))
    steps = [v[0] for k, v in rows.items() if "wgw_kernel" in k]
    steps = steps[0] if steps else 1
    return {k: (v[0] / steps, v[1] / steps / 1e3) for k, v in rows.items()}, steps


a, sa = load(sys.argv[1])
b, sb = load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
print("steps: %d | %d" % (sa, sb))
keys = sorted(set(a) | set(b), key=lambda k: -(max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1])))
ta = tb = la = lb = 0.0
for k in keys:
    ca, ua = a.get(k, (0, 0))
    cb, ub = b.get(k, (0, 0))
    ta += ua
    tb += ub
    la += ca
    lb += cb
    if abs(ua - ub) >= thr or (ca != cb and max(ua, ub) >= thr):
        print("%-78s %6.2f x %8.1f us | %6.2f x %8.1f us  %+8.1f" % (k[:78], ca, ua, cb, ub, ub - ua))
print("TOTAL kernel us/step %.1f | %.1f  (%+.1f)   launches/step %.1f | %.1f" % (ta, tb, tb - ta, la, lb))
