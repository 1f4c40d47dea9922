cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potrf or factor or chol or update_k or potri or flow" 2>&1 | tail -2 > gpurun_out/r48_tests.txt
GPC_LIB_VARIANT=pf3 python -m pytest tests/test_gpu_parity.py -x -q -k "potrf or factor or chol or update_k or potri or flow" 2>&1 | tail -2 >> gpurun_out/r48_tests.txt
for v in "" pf2 pf3; do
  echo "== variant '$v'" >> gpurun_out/r48_ab.txt
  GPC_LIB_VARIANT=$v python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r48_ab.txt
  GPC_LIB_VARIANT=$v python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r48_ab.txt
  GPC_LIB_VARIANT=$v GPC_PANEL_FLOW=1 GPC_PANEL_FLOW_TRACE=2 python tools/flow_check.py 1024 child /tmp/x.npy 2>/dev/null | head -34 | tail -33 >> gpurun_out/r48_ab.txt
done
GPC_LIB_VARIANT=pf3 python tools/flow_soak.py 30 2>&1 | tail -1 >> gpurun_out/r48_ab.txt
