#!/bin/bash
# Assembly of every kernel source (device only, the flags of faststyle_amd/build.py) into /tmp/fs_asm, then tools/waitcnt_scan.py over it.
#   tools/waitcnt_scan.sh [file.hip ...]     (default: every csrc/*.hip)
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/fs_asm
mkdir -p $O
F="$@"; [ -z "$F" ] && F=$(ls $R/faststyle_amd/csrc/*.hip)
for f in $F; do
  b=$(basename $f .hip); X=""; [ $b = fs_wgw ] && X="-fslp-vectorize"
  echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize $X -I $R/faststyle_amd/csrc -I $R/include --offload-device-only -S $f -o $O/$b.s 2>/dev/null"
done | xargs -P 6 -I{} sh -c "{}"
python $R/tools/waitcnt_scan.py $(for f in $F; do echo $O/$(basename $f .hip).s; done)
