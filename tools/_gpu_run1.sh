echo "=== tests"
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_path_parity.py -x -q -m gpu 2>&1 | tail -3
echo "=== bench"
timeout 300 python bench.py --steps 20 --warmup 3 --no-stylize 2>&1 | grep -v amdgpu.ids | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 3 --no-stylize 2>&1 | grep -v amdgpu.ids | cut -c1-200
echo "=== step trace"
timeout 300 python tools/step_trace.py 2>&1 | grep -v amdgpu.ids | head -40
