export TMPDIR=/tmp
O=gpurun_out/r5c9
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1
tail -5 $O/collect.log
timeout 600 python bench.py > $O/bench_default_run.json 2> $O/bench_default_run.err
tail -1 $O/bench_default_run.err | cut -c1-700
for L in routefused default2; do
  if [ "$L" = routefused ]; then export FS_VGG_ROUTE_FUSED=1; else unset FS_VGG_ROUTE_FUSED; fi
  timeout 300 python bench.py --no-cpu-baseline --no-stylize --steps 40 --b4-steps 100 > $O/bench_$L.json 2> $O/bench_$L.err
  echo "$L $(tail -1 $O/bench_$L.err | cut -c1-330)"
done
