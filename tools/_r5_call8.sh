export TMPDIR=/tmp
O=gpurun_out/r5c8
mkdir -p $O
for T in 1024 4096 16384 1000000; do
  echo "== FS_FINALIZE_MIN_T=$T"
  FS_FINALIZE_MIN_T=$T timeout 120 python tools/fwd720.py 720 1280 1 fp32 2>&1 | grep -v amdgpu.ids | tail -1
  FS_FINALIZE_MIN_T=$T timeout 120 python tools/fwd720.py 1080 1920 8 bf16 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/finalize_min_t.txt
for L in fork nofork fork2 nofork2; do
  if [ "${L:0:2}" = no ]; then export FS_NO_SIDE_STREAM=1; else unset FS_NO_SIDE_STREAM; fi
  timeout 300 python bench.py --no-cpu-baseline --no-stylize --no-b4 --steps 40 > $O/bench_$L.json 2> $O/bench_$L.err
  echo "$L $(tail -1 $O/bench_$L.err | cut -c1-120)"
done | tee $O/fork_ab.txt
