#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2f
rm -rf $O && mkdir -p $O
( time timeout 500 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -12 $O/bench.err; cut -c1-400 $O/bench.json
