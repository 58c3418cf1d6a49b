export TMPDIR=/tmp
O=gpurun_out/r5c6
mkdir -p $O
echo "== SLP everywhere + FS_W4_MFMA_SETTLE: the F(4x4) kernel tests and the round-4 reproducer"
FASTSTYLE_HIP_LIB=$PWD/exp/libslp.so timeout 600 python tools/w4_slp_repro.py 2>&1 | grep -v amdgpu.ids | head -30 | tee $O/slp_repro.txt
FASTSTYLE_HIP_LIB=$PWD/exp/libslp.so timeout 900 python -m pytest tests/test_kernels_parity.py -x -q -m gpu -k "f4x4 or wino4 or winograd" 2>&1 | tail -5 | tee $O/slp_pytest.txt
echo "== A/B: no SLP anywhere (product) against the round-4 flags (SLP off in the F(4x4) files only)"
for L in new r4 new2 r42; do
  if [ "${L:0:2}" = r4 ]; then export FASTSTYLE_HIP_LIB=$PWD/exp/libslp_r4.so; else unset FASTSTYLE_HIP_LIB; fi
  timeout 400 python bench.py --no-cpu-baseline --steps 30 --b4-steps 100 > $O/bench_$L.json 2> $O/bench_$L.err
  tail -1 $O/bench_$L.err | cut -c1-600
done
unset FASTSTYLE_HIP_LIB
