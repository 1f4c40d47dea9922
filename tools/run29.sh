cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "grad" 2>&1 | tail -3 > gpurun_out/r29_tests.txt
for D in 4 8 16 32; do python tools/grad_bench.py 65536 $D 2>/dev/null | grep "kern_grad rbfard"; done > gpurun_out/r29_grad.txt 2>&1
