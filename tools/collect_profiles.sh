#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box from the repo root; everything is written to
# gpurun_out/prof/ -- copy what should be judged into profiles/rNN_*).  Counters are collected in their own passes
# (--kernel-trace + --pmc only).  Every python command runs under its own timeout.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
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
# 1. per-kernel time of the train step at the metric's configuration (batch 32 on one GPU): hipGraph-replayed timed region
#    + the eager instrumented passes, both counted; no b4 leg / no inference extras in this process, so the per-kernel
#    averages are those of the b32 step
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_b32 --output-format csv -- python $R/bench.py --no-cpu-baseline --no-stylize --no-b4 --steps 30 > $O/bench_b32_under_rocprofv3.json 2> $O/bench_b32_stderr.txt
stats_table $O/stats_b32 $O/kernel_stats_b32.txt
#    ... and at batch 4 per GPU (BASELINE configs[2])
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_b4 --output-format csv -- python $R/bench.py --no-cpu-baseline --no-stylize --no-b4 --batch-per-gpu 4 --steps 100 > $O/bench_b4_under_rocprofv3.json 2> $O/bench_b4_stderr.txt
stats_table $O/stats_b4 $O/kernel_stats_b4.txt
#    ... and the launcher path on the one GPU: bench.py starts its own rank through torch.distributed.run (RCCL, world size 1)
timeout 300 python $R/bench.py --gpus 1 --launch --no-cpu-baseline --no-stylize --steps 30 > $O/bench_b32_selflaunch_world1.json 2> $O/bench_selflaunch_stderr.txt
# 2. HBM traffic counters of the b32 step, one pass each, eager launches so every launch is a separate dispatch record
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_b32_$C --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-stylize --no-b4 --no-graph > /dev/null 2>&1
done
# 3. inference: 720p fp32 batch 1 (configs[1]) and 1080p bf16 batch 8 (configs[4]): kernel stats + the same two counters
for CFG in "720 1280 1 fp32" "1080 1920 8 bf16"; do
  set -- $CFG
  T=${1}p_b${3}_${4}
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_$T --output-format csv -- python $R/tools/fwd720.py $1 $2 $3 $4 > $O/fwd_$T.log 2>&1
  stats_table $O/stats_$T $O/kernel_stats_$T.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_${T}_$C --output-format csv -- python $R/tools/fwd720.py $1 $2 $3 $4 > /dev/null 2>&1
  done
done
# 4. matrix-core utilisation counters of the b32 step (one pass; tools/pmc_mfma.py folds them per kernel symbol)
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $O/pmc_mfma --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-stylize --no-b4 --no-graph > /dev/null 2>&1
cd $R
python tools/pmc_mfma.py $O/pmc_mfma $O/mfma_util_b32.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES (one pass) on MI355X; python bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-stylize --no-b4 --no-graph (train step, batch 32); per-launch averages" > $O/mfma_util_b32.txt 2>&1
PROV="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on MI355X, per-launch average over all launches of the kernel. Counters are KiB; gfx950 correction (MI355X_MICROARCH.md HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled. traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."
python tools/pmc_traffic.py $O/pmc_b32_FETCH_SIZE $O/pmc_b32_WRITE_SIZE $O/hbm_traffic_pmc.json "$PROV Command: python bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-stylize --no-b4 --no-graph (train step, batch 32)." 32 > $O/traffic_b32.txt
python tools/pmc_traffic.py $O/pmc_720p_b1_fp32_FETCH_SIZE $O/pmc_720p_b1_fp32_WRITE_SIZE $O/hbm_traffic_720p_fp32.json "$PROV Command: python tools/fwd720.py 720 1280 1 fp32 (23 forward passes)." - 23 > $O/traffic_720p.txt
python tools/pmc_traffic.py $O/pmc_1080p_b8_bf16_FETCH_SIZE $O/pmc_1080p_b8_bf16_WRITE_SIZE $O/hbm_traffic_1080p_b8_bf16.json "$PROV Command: python tools/fwd720.py 1080 1920 8 bf16 (23 forward passes)." - 23 > $O/traffic_1080p.txt
find $O -type d \( -name 'stats_*' -o -name 'pmc_*' \) -prune -exec rm -rf {} +
head -14 $O/kernel_stats_b32.txt; head -8 $O/traffic_b32.txt; tail -1 $O/bench_b32_under_rocprofv3.json | cut -c1-200; cat $O/fwd_*.log | grep forward
