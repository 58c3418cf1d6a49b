export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p720 --output-format csv -- python $R/tools/fwd720.py 720 1280 1 2>&1 | grep -v amdgpu.ids | tail -2
cd $R
F=$(find gpurun_out/p720 -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $F 2>/dev/null | head -30 || head -30 $F
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:v for k,v in d.items() if 'fps' in k or 'extra' in k or k in ('value','ms_per_step')}); print(d.get('extras'))"
