export TMPDIR=/tmp
O=gpurun_out/r5c3
mkdir -p $O
timeout 1200 python -m pytest tests/test_generations.py tests/test_path_parity.py tests/test_golden_fixtures.py tests/test_abi_errors.py tests/test_scripts.py -x -q -m gpu -k "graph_replayed or perceptual or train_step or b32_256 or injected or abi or second_stream or forward_kernel_choices or golden or resume or train_py" > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for L in new base new2 base2; do
  if [ "${L:0:4}" = base ]; then export FASTSTYLE_HIP_LIB=$PWD/exp/libbase.so; else unset FASTSTYLE_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-stylize --steps 30 --b4-steps 100 > $O/bench_$L.json 2> $O/bench_$L.err
  tail -1 $O/bench_$L.err | cut -c1-400
done
unset FASTSTYLE_HIP_LIB
bash tools/kstats.sh r5c3 b32 b4
