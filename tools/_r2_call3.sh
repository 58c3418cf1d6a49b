#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r2c
rm -rf $O && mkdir -p $O
( time python -m pytest tests/test_rccl_world1.py tests/test_bench_contract.py tests/test_kernels_parity.py tests/test_abi.py -x -q -m gpu ) > $O/pytest.log 2>&1
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; tail -3 $O/bench.err; cut -c1-600 $O/bench.json
