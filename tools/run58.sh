cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "gplvm or lvm or host or cli or chol_inverse or oil" 2>&1 | tail -3 > gpurun_out/r58.txt
for i in 1 2 3 4 5; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "wall\|Log likelihood\|Final" | tail -2; done >> gpurun_out/r58.txt
