#!/usr/bin/env python
"""Static scan of gfx950 assembly for s_waitcnt vmcnt(k) instructions that DRAIN FRESHLY ISSUED LOADS inside a loop (round 6).

A software-pipelined kernel issues the loads of tile t + 1, multiplies tile t, then waits.  The wait-count insertion pass of the compiler is
conservative across control flow: loads issued (and waited for) inside conditional blocks stay "possibly pending" at a loop header, two arms of a
branch that load into the same registers are separated by vmcnt(0), a loop with two back-edge states takes the stricter one.  The result is a
s_waitcnt vmcnt(0..few) in the middle of an issue phase: the loads meant to travel beside the matrix instructions are waited for before the first of
them (fs_wino6.hip, fs_wgrad2.hip, fs_wgw.hip before round 6: 13 %, 5 % and 6 % of those kernels' time).

For every innermost-ish loop that holds matrix instructions this walks the body linearly, keeps the list of vector-memory loads issued so far and, at
every s_waitcnt vmcnt(k), finds the YOUNGEST load the wait forces to complete.  It reports the wait when fewer than --mfma matrix instructions (default
8) and fewer than --dist instructions (default 250) lie between that load and the wait -- i.e. a load is waited for almost as soon as it is issued.
Loads of the previous trip of the loop are modelled by walking the body twice.

usage: waitcnt_scan.py file.s [...]      (hipcc --offload-arch=gfx950 -O3 ... --offload-device-only -S x.hip -o x.s;  tools/waitcnt_scan.sh does all)"""
import argparse
import re
import subprocess


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def scan(path, min_mfma, max_dist, max_len):
    lines = open(path).read().split("\n")
    funcs = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    found = 0
    per_fn = {}
    for start, fn in funcs:
        end = next((i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        body = lines[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = {}
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)   # (conditional back edges only: an unconditional backward s_branch is an out-of-line block returning)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo = labels[m.group(1)]
                if lo not in loops or i < loops[lo]:
                    loops[lo] = i
        reported = set()
        for lo, hi in sorted(loops.items()):
            seg = [(j, body[j].strip()) for j in range(lo, hi + 1) if body[j].strip() and not body[j].strip().startswith(";")]
            if not any("v_mfma" in s for _, s in seg) or len(seg) > max_len:
                continue
            loads = []       # (position, mfma count at issue)
            pos = nm = 0
            for trip in range(2):
                for j, s in seg:
                    op = s.split()[0]
                    pos += 1
                    if op.startswith("v_mfma"):
                        nm += 1
                    elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                        loads.append((pos, nm))
                    elif op == "s_waitcnt":
                        m = re.search(r"vmcnt\((\d+)\)", s)
                        if m and trip == 1:
                            k = int(m.group(1))
                            if len(loads) > k:
                                p, n0 = loads[-(k + 1)]
                                if nm - n0 < min_mfma and pos - p < max_dist and j not in reported:
                                    reported.add(j)
                                    if not found:
                                        print("== " + path)
                                    found += 1
                                    per_fn.setdefault(fn, []).append("+%d: vmcnt(%d) <- load %d instr / %d mfma earlier" % (j, k, pos - p, nm - n0))
                        if m:
                            k = int(m.group(1))
                            loads = loads[len(loads) - k:] if k else []
    for fn, hits in per_fn.items():
        print("  %-120s %3d: %s%s" % (demangle(fn)[:120], len(hits), "; ".join(hits[:4]), " ..." if len(hits) > 4 else ""))
    return found


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--mfma", type=int, default=8)
    ap.add_argument("--dist", type=int, default=250)
    ap.add_argument("--maxlen", type=int, default=4000)
    a = ap.parse_args()
    total = sum(scan(f, a.mfma, a.dist, a.maxlen) for f in a.files)
    print("%d suspicious wait(s)" % total)
