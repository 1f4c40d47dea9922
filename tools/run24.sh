cd $GRAFT_REPO_ROOT
GPC_GPLVM_TIMING=1 gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model > gpurun_out/r24_gplvm_timing.txt 2>&1
bash tools/trace.sh gplvm gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model > gpurun_out/r24_gplvm_trace.txt 2>&1
