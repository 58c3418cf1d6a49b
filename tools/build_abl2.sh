#!/bin/bash
# tools/build_abl2.sh <source.hip> <macro> <values...>: timing builds of one translation unit with -D<macro>=<v> linked against the other (current) objects -> exp/libabl_<v>.so
F=$1; M=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$(basename $F .hip)
OBJS=$(ls $R/faststyle_amd/build/*.o | grep -v "/$B.hip.o")
mkdir -p $R/exp
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize -D$M=$v -I $R/faststyle_amd/csrc -I $R/include -c $R/faststyle_amd/csrc/$B.hip -o $R/exp/${B}_$v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/exp/${B}_$v.o -o $R/exp/libabl_$v.so && echo built $v ) &
done
wait
