#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2d
rm -rf $O && mkdir -p $O
nproc > $O/cpu.log; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))" >> $O/cpu.log
( time timeout 400 python -c "
import bench, json
print(json.dumps(bench.cpu_baseline(256)))
" ) > $O/cpu.json 2> $O/cpu.err
cat $O/cpu.log; tail -12 $O/cpu.err
