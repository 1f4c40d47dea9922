cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_dtc.py -x -q -k "grad or pair_walk or gplvm or dtc or synthetic or fixture or goldens" 2>&1 | tail -4 > gpurun_out/r14_tests.txt
for D in 8 16 32; do python tools/grad_bench.py 65536 $D 2>/dev/null | grep "kern_grad "; done > gpurun_out/r14_grad.txt 2>&1
for D in 8 16 32; do python tools/grad_bench.py 32768 $D 2>/dev/null | grep "kern_gradx"; done >> gpurun_out/r14_grad.txt 2>&1
