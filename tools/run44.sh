cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potri" 2>&1 | tail -3 > gpurun_out/r44_tests.txt
for B in 0 1024 2048 4096; do
  echo "GPC_POTRI_BAND=$B" >> gpurun_out/r44.txt
  for N in 6144 8192 12288 16384 20480; do
    GPC_POTRI_BAND=$B python tools/potri_bench.py $N 2>/dev/null | tail -1 | cut -c1-70 >> gpurun_out/r44.txt
  done
done
