export TMPDIR=/tmp
O=gpurun_out/r5c13
mkdir -p $O
timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1
tail -3 $O/collect.log
for f in kernel_stats_b32.txt kernel_stats_b4.txt kernel_stats_720p_b1_fp32.txt kernel_stats_1080p_b8_bf16.txt hbm_traffic_pmc.json hbm_traffic_720p_fp32.json hbm_traffic_1080p_b8_bf16.json mfma_util_b32.json bench_b32_under_rocprofv3.json bench_b4_under_rocprofv3.json bench_b32_selflaunch_world1.json; do cp gpurun_out/prof/$f profiles/r05_$f; done
timeout 600 python bench.py > $O/bench_default_run.json 2> $O/bench_default_run.err
tail -1 $O/bench_default_run.err | cut -c1-700
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_path_parity.py tests/test_generations.py -x -q -m gpu -k "filter_gradient or partial_sums or train_step or graph_replayed or b32_256" 2>&1 | tail -2
