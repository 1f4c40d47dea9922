cd $GRAFT_REPO_ROOT
for L in 0 1; do
  echo "GPC_PANEL_FLOW_LEAN=$L" >> gpurun_out/r39.txt
  GPC_PANEL_FLOW_LEAN=$L python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r39.txt
  GPC_PANEL_FLOW_LEAN=$L python tools/factor_sweep.py 1000 2048 4096 8192 16384 2>/dev/null >> gpurun_out/r39.txt
done
bash tools/run23.sh
cat gpurun_out/r23_gplvm.txt >> gpurun_out/r39.txt
