cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potri or gemm or trsm or posterior" 2>&1 | tail -2 > gpurun_out/r50_tests.txt
for G in 16 64; do
  echo "== GPC_GEMM_DEAL_GROUP=$G" >> gpurun_out/r50.txt
  for N in 5120 6144 8192 12288 16384 20480; do
    GPC_GEMM_DEAL_GROUP=$G python tools/potri_bench.py $N 2>/dev/null | tail -1 | cut -c1-75 >> gpurun_out/r50.txt
  done
done
