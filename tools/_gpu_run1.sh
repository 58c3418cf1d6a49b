echo "=== trace"
FASTSTYLE_HIP_LIB=exp/libtrace.so timeout 300 python tools/conv_trace.py vgg1_2_n4 vgg3_2_n4 res_n4 2>&1 | grep -v "amdgpu.ids\|HW_ID\|earliest\|start offsets"
echo "=== micro"
ITERS=30 timeout 300 python tools/micro_conv.py vgg1_2_n4 vgg2_2_n4 vgg3_2_n4 vgg4_2_n4 vgg4_1_n4 res_n4 res_720p 2>&1 | grep -v amdgpu.ids
echo "=== tests"
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_path_parity.py -x -q -m gpu 2>&1 | tail -3
echo "=== bench"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:v for k,v in d.items() if 'fps' in k or k in ('value','ms_per_step')}); print(d['roofline']['per_kernel'])"
