cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "potri or chol_inverse" 2>&1 | tail -5 > gpurun_out/r3_tests.txt
python -m pytest tests/test_grid_gpu.py -x -q -k "rehearsal" 2>&1 | tail -8 >> gpurun_out/r3_tests.txt
python tools/potri_inplace_ab.py 16384 24576 32768 65536 > gpurun_out/r3_potri.txt 2>&1
