#!/bin/bash
# round-2 first look: per-kernel time of the train step at batch 32 on one GPU, and of the 720p / 1080p-bf16 forward
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2a
rm -rf $O && mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/b32 --output-format csv -- python $R/bench.py --batch-per-gpu 32 --steps 6 --warmup 2 --no-cpu-baseline --no-stylize > $O/b32.json 2> $O/b32.err
rocprofv3 --kernel-trace --stats -d $O/f720 --output-format csv -- python $R/tools/fwd720.py 720 1280 1 > $O/f720.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/f1080 --output-format csv -- python $R/tools/fwd720.py 1080 1920 8 bf16 > $O/f1080.log 2>&1
cd $R
for d in b32 f720 f1080; do
  F=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  python - "$F" > $O/$d.stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-100s %7s %14s %12s %7s" % ("Name", "Calls", "TotalDur(ns)", "AvgDur(ns)", "Pct"))
for r in rows:
    print("%-100s %7s %14s %12.0f %6.2f%%" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
PY
  find $O/$d -type f ! -name "*kernel_stats.csv" -delete
done
tail -3 $O/*.log; tail -1 $O/b32.json | cut -c1-400
