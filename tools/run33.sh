cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potrf or factor or chol or update_k or potri or trsm or flow or posterior" 2>&1 | tail -3 > gpurun_out/r33_tests.txt
for v in "" split0; do
  echo "== variant '$v'" >> gpurun_out/r33_ab.txt
  GPC_LIB_VARIANT=$v python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r33_ab.txt
  GPC_LIB_VARIANT=$v python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r33_ab.txt
  GPC_LIB_VARIANT=$v GPC_PANEL_FLOW=1 GPC_PANEL_FLOW_TRACE=2 python tools/flow_check.py 1024 child /tmp/x.npy 2>/dev/null | head -24 >> gpurun_out/r33_ab.txt
  GPC_LIB_VARIANT=$v python tools/potri_bench.py 8192 2>/dev/null | tail -1 >> gpurun_out/r33_ab.txt
  GPC_LIB_VARIANT=$v python tools/potri_bench.py 4096 2>/dev/null | tail -1 >> gpurun_out/r33_ab.txt
done
python tools/flow_soak.py 30 2>&1 | tail -3 >> gpurun_out/r33_ab.txt
