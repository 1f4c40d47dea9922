cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "grad" 2>&1 | tail -2 > gpurun_out/r56.txt
for v in base ""; do
  echo "== variant '$v'" >> gpurun_out/r56.txt
  for N in 500 1000 1500; do for D in 4 12; do GPC_LIB_VARIANT=$v python tools/grad_bench.py $N $D 2>/dev/null | grep "kern_grad " >> gpurun_out/r56.txt; done; done
done
echo "== gplvm new lib" >> gpurun_out/r56.txt
for i in 1 2 3 4; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "wall" ; done >> gpurun_out/r56.txt
cp gpc_amd/lib/libgpc_hip.so /tmp/new.so; cp gpc_amd/lib/libgpc_hip_base.so gpc_amd/lib/libgpc_hip.so
echo "== gplvm base lib" >> gpurun_out/r56.txt
for i in 1 2 3 4; do gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "wall" ; done >> gpurun_out/r56.txt
cp /tmp/new.so gpc_amd/lib/libgpc_hip.so
