export TMPDIR=/tmp
O=gpurun_out/r5c11
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 600 python -m pytest tests/test_path_parity.py -q -s -m gpu -k "train_step_256_matches or b32_256" 2>&1 | grep -E "data-parallel identity|b32 256x256|passed|failed" | tee $O/dp_values.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.txt
timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1
tail -3 $O/collect.log
R=$PWD
for L in new r4 new2 r42; do
  if [ "${L:0:2}" = r4 ]; then cd $R/exp/r4tree; else cd $R; fi
  timeout 400 python bench.py --no-cpu-baseline --steps 30 --b4-steps 100 > $R/$O/ab_$L.json 2> $R/$O/ab_$L.err
  cd $R
  python - $O/ab_$L.json $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-5s b32 %8.2f img/s %7.3f ms  b4 %8.2f img/s %6.3f ms  720p %7.1f  1080p bf16 %7.1f  fp32 %6.1f" % (sys.argv[2], d["value"], d["ms_per_step"], d["train_b4_per_gpu"]["images_per_sec"], d["train_b4_per_gpu"]["ms_per_step"], d["stylize_720p"]["fps"], d["stylize_1080p_b8_bf16"]["fps"], d["stylize_1080p_b8_fp32"]["fps"]))
PY
done | tee $O/ab_r4_vs_r5.txt
