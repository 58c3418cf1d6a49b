echo "=== bf16 tests"
timeout 900 python -m pytest tests/test_path_parity.py -x -q -m gpu -k "bf16 or 1080" 2>&1 | tail -3
echo "=== fwd"
timeout 300 python tools/fwd720.py 1080 1920 8 bf16 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/fwd720.py 1080 1920 8 bf16 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/fwd720.py 720 1280 1 bf16 2>&1 | grep -v amdgpu.ids
