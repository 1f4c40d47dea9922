cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py -x -q -k "grad or gram or gplvm" 2>&1 | tail -3 > gpurun_out/r25_tests.txt
for i in 1 2 3; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "wall time" | tail -1; done >> gpurun_out/r25_tests.txt 2>&1
GPC_GPLVM_TIMING=1 gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep "gplvm timing" >> gpurun_out/r25_tests.txt
