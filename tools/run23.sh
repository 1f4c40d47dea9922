cd $GRAFT_REPO_ROOT
for i in 1 2 3; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "evaluations\|seconds\|took" | tail -3; done > gpurun_out/r23_gplvm.txt 2>&1
