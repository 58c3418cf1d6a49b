#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box from the repo root; writes to gpurun_out/
# first, copy what should be judged into profiles/).  Counters are collected in their own passes (kernel-trace only).
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd /tmp
# 1. per-kernel time of the bench command restricted to the training step (--no-stylize: otherwise the 720p/1080p
#    inference extras launch the same conv kernels at other shapes and the per-kernel averages are no longer those of
#    the step; hipGraph-replayed timed region + the eager instrumented pass, both counted)
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python $R/bench.py --no-cpu-baseline --no-stylize > $O/bench_under_rocprofv3.json 2> $O/bench_stderr.txt
# 2. HBM traffic counters, one pass each, eager launches so every launch is a separate dispatch record
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stylize --no-graph > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stylize --no-graph > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/fetch $O/write $O/hbm_traffic_pmc.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) around \`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stylize --no-graph\` on MI355X; per-launch average over all launches of the kernel. Counters are KiB; gfx950 correction (MI355X_MICROARCH.md HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled (check: maxpool_kernel reads its input once and writes a quarter of it). traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."
F=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python - "$F" > $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-100s %7s %14s %12s %7s" % ("Name", "Calls", "TotalDur(ns)", "AvgDur(ns)", "Pct"))
for r in rows:
    print("%-100s %7s %14s %12.0f %6.2f%%" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
PY
head -30 $O/kernel_stats.txt
tail -1 $O/bench_under_rocprofv3.json | cut -c1-300
