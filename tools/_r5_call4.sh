export TMPDIR=/tmp
O=gpurun_out/r5c4
mkdir -p $O
for V in "8 0" "4 0" "8 4" "8 8"; do
  set -- $V
  echo "== FS_INBWD_PUNR=$1 FS_INBWD_AUNR=$2"
  FS_INBWD_PUNR=$1 FS_INBWD_AUNR=$2 timeout 300 python tools/micro_inbwd.py res_b32 res82_b32 res_b4 init0_b32 init1_b32 up0_b32 up1_b32 init0_b4 2>&1 | grep -v amdgpu.ids
done | tee $O/micro_inbwd.txt
