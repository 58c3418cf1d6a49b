echo "=== wgrad trace"
FASTSTYLE_HIP_LIB=exp/libtrace.so timeout 300 python tools/wgrad_trace.py res_n4 s2_16_32 fold_like gram1_2 2>&1 | grep -v amdgpu.ids | grep -v "barrier\|wave-slot"
echo "=== micro wgrad"
ITERS=30 timeout 300 python tools/micro_wgrad.py 2>&1 | grep -v amdgpu.ids
echo "=== tests"
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_path_parity.py -x -q -m gpu 2>&1 | tail -3
echo "=== bench"
timeout 300 python bench.py --steps 20 --warmup 3 --no-stylize 2>&1 | grep -v amdgpu.ids | cut -c1-200
