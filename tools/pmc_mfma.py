#!/usr/bin/env python
"""Fold one rocprofv3 PMC pass (--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ..., CSV output) into
a per-kernel matrix-core utilisation table: what BASELINE.json's north_star calls "MFMA-utilisation counters against gfx950 peak".

    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDS x SIMDs)      SIMDs = 256 CUs x 4, XCDS = 8

(rocprofv3's own derived MfmaUtil = reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) / (reduce(GRBM_GUI_ACTIVE,max) x SIMD_NUM); its CSV gives
each counter already SUMMED over its dimensions, and every XCD has its own GRBM, so the per-launch GRBM_GUI_ACTIVE value is 8 x
the busy cycles of one XCD -- checked on wino4_conv_kernel: busy / 1024 SIMDs = 655k cycles per SIMD = the launch's executed
FLOPs / 64 flop per cycle per SIMD, and GUI_ACTIVE / 8 / launch duration = 2.36 GHz.)

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over every SIMD of the chip
(MI355X_MICROARCH.md, per-instruction constants: 32 per v_mfma_f32_16x16x4_f32, 64 per v_mfma_f32_32x32x2_f32); GRBM_GUI_ACTIVE the
cycles the launch kept the GPU busy.  The ratio is clock-independent -- unlike TFLOP/s against the 2.4 GHz peak, it does not
charge the kernel for the clock the part sustains under fp32 matrix load.

usage: pmc_mfma.py <dir> <out.json> [provenance text]"""
import csv
import glob
import json
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402

SIMDS = 256 * 4
XCDS = 8


def main():
    d, out = sys.argv[1:3]
    prov = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = {}
    for k, c in acc.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        n = len(c["GRBM_GUI_ACTIVE"])
        busy, active = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n, sum(c["GRBM_GUI_ACTIVE"]) / n
        rows[k] = {"launches_sampled": n, "SQ_VALU_MFMA_BUSY_CYCLES": round(busy), "GRBM_GUI_ACTIVE": round(active),
                   "mfma_util": round(busy / (active / XCDS * SIMDS), 4) if active else None}
        for extra in ("SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA"):
            if extra in c:
                rows[k][extra] = round(sum(c[extra]) / len(c[extra]))
    rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"] * kv[1]["launches_sampled"]))
    json.dump({"_provenance": prov, "simds": SIMDS, "xcds": XCDS, "kernels": rows}, open(out, "w"), indent=1)
    for k, v in list(rows.items())[:16]:
        print("%-44s %4d launches  GUI_ACTIVE %10d  MFMA busy %13d  util %s" % (k, v["launches_sampled"], v["GRBM_GUI_ACTIVE"], v["SQ_VALU_MFMA_BUSY_CYCLES"], v["mfma_util"]))


if __name__ == "__main__":
    main()
