export TMPDIR=/tmp
O=gpurun_out/r5c12
mkdir -p $O
for L in default nofuse wgwslp default2 nofuse2 wgwslp2; do
  unset FASTSTYLE_HIP_LIB FS_INBWD_FUSED
  case $L in nofuse*) export FS_INBWD_FUSED=0;; wgwslp*) export FASTSTYLE_HIP_LIB=$PWD/exp/libwgwslp.so;; esac
  timeout 300 python bench.py --no-cpu-baseline --no-stylize --steps 40 --b4-steps 150 > $O/b_$L.json 2> $O/b_$L.err
  python - $O/b_$L.json $L <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pk = d["roofline"]["per_kernel"]
print("%-9s b32 %8.2f img/s %7.3f ms  b4 %8.2f img/s %6.3f ms | wgw %.3f  tnet-res %.3f ms" % (sys.argv[2], d["value"], d["ms_per_step"], d["train_b4_per_gpu"]["images_per_sec"], d["train_b4_per_gpu"]["ms_per_step"], pk["wgw_kernel"]["ms_per_step"], pk["wino4t_conv_kernel (transform-net residual convs)"]["ms_per_step"]))
PY
done | tee $O/ab.txt
