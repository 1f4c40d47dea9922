cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_dtc.py tests/test_host_layer.py -x -q -k "exp_primitive or grad or goldens or synthetic or dtc or gplvm or fixtures" 2>&1 | tail -3 > gpurun_out/r21_tests.txt
for D in 4 8 16 32; do python tools/grad_bench.py 65536 $D 2>/dev/null | grep "kern_grad "; done > gpurun_out/r21_grad.txt 2>&1
