export TMPDIR=/tmp
O=gpurun_out/r5c10
mkdir -p $O
for V in "96 8" "48 8" "192 8" "384 8" "192 4" "768 8"; do
  set -- $V
  echo "== FS_INBWD_REC_MAXT=$1 FS_INBWD_PUNR=$2"
  FS_INBWD_REC_MAXT=$1 FS_INBWD_PUNR=$2 ITERS=16 timeout 200 python tools/micro_inbwd.py init0_b32 init1_b32 up1_b32 up0_b32 init0_b4 2>&1 | grep -v amdgpu.ids | cut -c1-120
done | tee $O/micro_maxt.txt
