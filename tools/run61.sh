cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python -m pytest tests/test_dtc.py -x -q -m gpu -k "unwritten" 2>&1 | tail -1; done > gpurun_out/r61.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "defer" 2>&1 | tail -3 >> gpurun_out/r61.txt
