cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or syrk or potri or chol_inverse or trsm" 2>&1 | tail -5 > gpurun_out/r2_tests.txt
python -m pytest tests/test_grid_gpu.py tests/test_host_layer.py -x -q -k "rehearsal or distinct or rccl" 2>&1 | tail -8 >> gpurun_out/r2_tests.txt
GPC_POTRI_INPLACE_MINN=2048 bash tools/trace.sh potri python tools/potri_only.py 65536 > gpurun_out/r2_trace_potri.txt 2>&1
