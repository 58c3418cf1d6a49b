#!/bin/bash
# tools/build_abl.sh <macro> <values...>: timing builds of fs_wino6.hip with -D<macro>=<v> linked against the other (current) objects -> exp/libabl_<v>.so
M=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OBJS=$(ls $R/faststyle_amd/build/*.o | grep -v fs_wino6)
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize -D$M=$v -I $R/faststyle_amd/csrc -c $R/faststyle_amd/csrc/fs_wino6.hip -o $R/exp/w6_$v.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/exp/w6_$v.o -o $R/exp/libabl_$v.so && echo built $v ) &
done
wait
