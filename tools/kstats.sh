#!/bin/bash
# rocprofv3 --kernel-trace --stats of the train step at batch 32 and batch 4 (the first two steps of tools/collect_profiles.sh), into gpurun_out/<tag>/
#   gpurun -- 'bash tools/kstats.sh <tag> [b32] [b4]'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
shift
cd /tmp
stats_table() {   # <dir with *kernel_stats.csv> <out.txt>
python - "$(find $1 -name '*kernel_stats.csv' | head -1)" > $2 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-100s %7s %14s %12s %7s" % ("Name", "Calls", "TotalDur(ns)", "AvgDur(ns)", "Pct"))
for r in rows:
    print("%-100s %7s %14s %12.0f %6.2f%%" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
PY
}
for W in "$@"; do
  case $W in
    b32) timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_b32 --output-format csv -- python $R/bench.py --no-cpu-baseline --no-stylize --no-b4 --steps 30 > $O/bench_b32_under_rocprofv3.json 2> $O/bench_b32_stderr.txt
         stats_table $O/stats_b32 $O/kernel_stats_b32.txt; rm -rf $O/stats_b32;;
    b4)  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_b4 --output-format csv -- python $R/bench.py --no-cpu-baseline --no-stylize --no-b4 --batch-per-gpu 4 --steps 100 > $O/bench_b4_under_rocprofv3.json 2> $O/bench_b4_stderr.txt
         stats_table $O/stats_b4 $O/kernel_stats_b4.txt; rm -rf $O/stats_b4;;
    720p) timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_720p --output-format csv -- python $R/tools/fwd720.py 720 1280 1 fp32 > $O/fwd_720p.log 2>&1
         stats_table $O/stats_720p $O/kernel_stats_720p_b1_fp32.txt; rm -rf $O/stats_720p;;
    1080p) timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_1080p --output-format csv -- python $R/tools/fwd720.py 1080 1920 8 bf16 > $O/fwd_1080p.log 2>&1
         stats_table $O/stats_1080p $O/kernel_stats_1080p_b8_bf16.txt; rm -rf $O/stats_1080p;;
  esac
done
