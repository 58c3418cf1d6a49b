for v in -1 0 3 4; do
  echo "=== variant $v"
  FS_CONV_VARIANT=$v ITERS=30 timeout 300 python tools/micro_conv.py res_n4 vgg4_2_n4 vgg4_1_n4 vgg3_2_n4 2>&1 | grep -v amdgpu.ids
  FS_CONV_VARIANT=$v STATS=1 ITERS=30 timeout 300 python tools/micro_conv.py res_n4 2>&1 | grep -v amdgpu.ids
done
for m in 256 384; do
  echo "=== min_wgs $m"
  FS_CONV_MIN_WGS=$m ITERS=30 timeout 300 python tools/micro_conv.py res_n4 vgg4_2_n4 vgg4_1_n4 vgg3_2_n4 2>&1 | grep -v amdgpu.ids
  FS_CONV_MIN_WGS=$m timeout 300 python bench.py --steps 20 --warmup 3 --no-stylize 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
