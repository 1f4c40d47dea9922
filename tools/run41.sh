cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "trsv or alpha or trsm or solve or posterior or loglik or cfg2" 2>&1 | tail -3 > gpurun_out/r41_tests.txt
for v in "" occ1 trsvold; do
  echo "== variant '$v'" >> gpurun_out/r41.txt
  for N in 1000 8192 32768 65536; do
    GPC_LIB_VARIANT=$v python tools/trsv_bench.py $N 1 2>/dev/null >> gpurun_out/r41.txt
  done
  GPC_LIB_VARIANT=$v python tools/trsv_bench.py 65536 3 2>/dev/null >> gpurun_out/r41.txt
done
