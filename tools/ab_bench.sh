#!/bin/bash
# Same-lease A/B of the working tree's library against the library of another commit (the method behind every "measured"
# claim of DESIGN.md since the second half of round 3: two alternations on ONE box, per kernel row, never a single pair of runs
# on different leases -- boxes differ by 2-4 %).
#   here (CPU):  tools/ab_bench.sh build <commit>     -> exp/libbase.so from that commit's csrc (exp/ is git-ignored but travels)
#   on the GPU:  gpurun -- 'bash tools/ab_bench.sh run'  -> bench.py with both libraries, new / base / new / base
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  T=$(mktemp -d)
  git -C "$R" archive "$2" faststyle_amd/csrc include | tar -x -C "$T"
  for f in "$T"/faststyle_amd/csrc/*.hip; do
    X=""; case "$(basename $f)" in fs_wino4*) X="-fno-slp-vectorize";; esac   # (faststyle_amd/build.py FILE_FLAGS)
    echo "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $X -c $f -o $T/$(basename $f).o"
  done | xargs -P 8 -I{} sh -c "{}"
  mkdir -p "$R/exp"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$T"/*.o -o "$R/exp/libbase.so"
  rm -rf "$T"
  echo "built exp/libbase.so from $2"
  exit 0
fi
export TMPDIR=/tmp
O=$R/gpurun_out/ab
rm -rf "$O" && mkdir -p "$O"
for L in new base new2 base2; do
  if [ "${L:0:4}" = base ]; then export FASTSTYLE_HIP_LIB=$R/exp/libbase.so; else unset FASTSTYLE_HIP_LIB; fi
  timeout 400 python "$R/bench.py" --no-cpu-baseline --steps 30 --b4-steps 100 > "$O/bench_$L.json" 2> "$O/bench_$L.err"
done
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
d = {L: json.loads(open(f"{o}/bench_{L}.json").read().strip().splitlines()[-1]) for L in ("new", "base", "new2", "base2")}
for L in d:
    print("%-6s b32 %8.2f img/s %7.3f ms  b4 %7.2f  720p %7.1f  1080p bf16 %7.1f  fp32 %6.1f" % (
        L, d[L]["value"], d[L]["ms_per_step"], d[L]["train_b4_per_gpu"]["images_per_sec"], d[L]["stylize_720p"]["fps"],
        d[L]["stylize_1080p_b8_bf16"]["fps"], d[L]["stylize_1080p_b8_fp32"]["fps"]))
pk = {L: d[L]["roofline"]["per_kernel"] for L in d}
for k in dict.fromkeys(list(pk["new"]) + list(pk["base"])):
    a = [pk[L].get(k, {}).get("ms_per_step", 0.0) for L in ("new", "new2")]
    b = [pk[L].get(k, {}).get("ms_per_step", 0.0) for L in ("base", "base2")]
    print("  %-62s new %7.3f %7.3f  base %7.3f %7.3f  %+7.3f ms/step" % (k, a[0], a[1], b[0], b[1], sum(a) / 2 - sum(b) / 2))
PY
