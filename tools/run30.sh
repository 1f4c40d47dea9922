cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potrf or factor or chol or update_k or potri" 2>&1 | tail -3 > gpurun_out/r30_tests.txt
for v in "" chol8; do
  echo "== variant '$v'" >> gpurun_out/r30_ab.txt
  GPC_LIB_VARIANT=$v python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r30_ab.txt
  GPC_LIB_VARIANT=$v python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r30_ab.txt
  GPC_LIB_VARIANT=$v GPC_PANEL_FLOW=1 GPC_PANEL_FLOW_TRACE=2 python tools/flow_check.py 1024 child /tmp/x.npy 2>/dev/null >> gpurun_out/r30_ab.txt
done
python tools/potri_bench.py 8192 2>/dev/null | tail -3 >> gpurun_out/r30_ab.txt
GPC_LIB_VARIANT=chol8 python tools/potri_bench.py 8192 2>/dev/null | tail -3 >> gpurun_out/r30_ab.txt
