export TMPDIR=/tmp
O=gpurun_out/r5c1
mkdir -p $O
timeout 900 python -m pytest tests/test_abi_errors.py tests/test_path_parity.py tests/test_generations.py -x -q -m gpu -k "abi or 16tile_f4x4 or instnorm or graph_replayed or train_step_256 or b32_256 or injected or second_stream" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for L in new old new2 old2; do
  if [ "${L:0:3}" = old ]; then export FS_INBWD_REC=0 FS_INBWD_FUSED=0; else unset FS_INBWD_REC FS_INBWD_FUSED; fi
  timeout 300 python bench.py --no-cpu-baseline --no-stylize --steps 30 --b4-steps 100 > $O/bench_$L.json 2> $O/bench_$L.err
  python - $O/bench_$L.json $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-5s b32 %8.2f img/s %7.3f ms   b4 %8.2f img/s %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["train_b4_per_gpu"]["images_per_sec"], d["train_b4_per_gpu"].get("ms_per_step")))
PY
done
