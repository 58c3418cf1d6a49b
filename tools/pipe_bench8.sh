#!/bin/bash
# Eight input pipelines at once on the GPU box's host (one per rank of an 8-GPU node; here they share the box's one GPU for the resize kernels):
# eight tools/pipe_bench.py processes, each pinned to its own eighth of the host's cores, started together; prints every pool's rate and the sum.
#   gpurun -- 'bash tools/pipe_bench8.sh [images per pool] [decode threads per pool]'
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-6000}
NC=$(nproc)
PER=$((NC / 8))
TH=${2:-$((PER < 24 ? PER : 24))}
O=$R/gpurun_out/pipe8
rm -rf $O && mkdir -p $O
echo "host cores $NC, $PER per pool, $TH decode threads per pool, $N images per pool"
for r in 0 1 2 3 4 5 6 7; do
  lo=$((r * PER)); hi=$((lo + PER - 1))
  taskset -c $lo-$hi timeout 600 python $R/tools/pipe_bench.py $N $TH > $O/pool$r.txt 2>&1 &
done
wait
grep -h pipeline $O/pool*.txt
grep -h pipeline $O/pool*.txt | awk '{s += $2} END {printf "aggregate: %.0f images/s over %d pools\n", s, NR}'
# the host half alone (framing + parsing + JPEG decode, no GPU), eight pools at once: what the box's cores sustain when every rank has a GPU of its own
for r in 0 1 2 3 4 5 6 7; do
  lo=$((r * PER)); hi=$((lo + PER - 1))
  taskset -c $lo-$hi timeout 600 python $R/tools/pipe_bench.py $N $TH host > $O/host$r.txt 2>&1 &
done
wait
grep -h pipeline $O/host*.txt | awk '{s += $2} END {printf "host half, eight pools at once: %.0f images/s aggregate (%.0f per pool)\n", s, s / NR}'
# one pool alone on the same cores, for the ratio
taskset -c 0-$((PER - 1)) timeout 600 python $R/tools/pipe_bench.py $N $TH 2>&1 | grep pipeline | sed 's/^/alone: /'
taskset -c 0-$((PER - 1)) timeout 600 python $R/tools/pipe_bench.py $N $TH host 2>&1 | grep pipeline | sed 's/^/alone: /'
