cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py -x -q -k "loglik or alpha or logdet or cfg2 or gplvm or host or chol_inverse or grad or posterior" 2>&1 | tail -2 > gpurun_out/r57.txt
for G in 0 1 0 1; do
  echo "== GPC_HOST_GATHER=$G" >> gpurun_out/r57.txt
  for i in 1 2 3; do GPC_HOST_GATHER=$G gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 tests/golden/oilTrain.svml /tmp/oil.model 2>&1 | grep -i "wall" ; done >> gpurun_out/r57.txt
done
GPC_HOST_GATHER=0 python tools/grad_bench.py 1000 4 2>/dev/null | grep "kern_grad " >> gpurun_out/r57.txt
GPC_HOST_GATHER=1 python tools/grad_bench.py 1000 4 2>/dev/null | grep "kern_grad " >> gpurun_out/r57.txt
