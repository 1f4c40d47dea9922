cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "potrf or factor or chol or update_k or potri or trsm or flow or posterior" 2>&1 | tail -3 > gpurun_out/r42_tests.txt
python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null > gpurun_out/r42_ab.txt
python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r42_ab.txt
GPC_PANEL_FLOW=1 GPC_PANEL_FLOW_TRACE=2 python tools/flow_check.py 1024 child /tmp/x.npy 2>/dev/null | head -36 >> gpurun_out/r42_ab.txt
python tools/potri_bench.py 8192 2>/dev/null | tail -1 >> gpurun_out/r42_ab.txt
python tools/potri_bench.py 4096 2>/dev/null | tail -1 >> gpurun_out/r42_ab.txt
python tools/flow_soak.py 30 2>&1 | tail -1 >> gpurun_out/r42_ab.txt
