cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potri or gemm or trsm or posterior" 2>&1 | tail -2 > gpurun_out/r49_tests.txt
for v in "" rr; do
  echo "== variant '$v'" >> gpurun_out/r49.txt
  for N in 6144 8192 12288 16384; do
    GPC_LIB_VARIANT=$v python tools/potri_bench.py $N 2>/dev/null | tail -1 | cut -c1-75 >> gpurun_out/r49.txt
  done
  GPC_LIB_VARIANT=$v python tools/potri_only.py 32768 > /dev/null 2>&1
  GPC_LIB_VARIANT=$v python tools/potri_inplace_ab.py 32768 2>/dev/null | tail -3 >> gpurun_out/r49.txt
done
