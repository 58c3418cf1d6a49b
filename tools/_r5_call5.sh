export TMPDIR=/tmp
O=gpurun_out/r5c5
mkdir -p $O
for V in 1024 2048 512; do
  echo "== FS_INBWD_APPLY_WGS=$V"
  FS_INBWD_APPLY_WGS=$V timeout 300 python tools/micro_inbwd.py res_b32 res82_b32 res_b4 init0_b32 init1_b32 up0_b32 up1_b32 init0_b4 2>&1 | grep -v amdgpu.ids
done | tee $O/micro_inbwd.txt
