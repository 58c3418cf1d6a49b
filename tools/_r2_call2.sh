#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2b
rm -rf $O && mkdir -p $O
cd /tmp
for B in 32 4; do
rocprofv3 --kernel-trace -d $O/t$B -- python $R/bench.py --batch-per-gpu $B --steps 3 --warmup 2 --no-cpu-baseline --no-stylize --no-graph > $O/t$B.json 2> $O/t$B.err
DB=$(find $O/t$B -name "*.db" | head -1)
python $R/tools/step_trace.py $DB $O/step_b$B.txt > /dev/null
rm -rf $O/t$B
done
