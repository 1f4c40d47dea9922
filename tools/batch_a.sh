cd $GRAFT_REPO_ROOT
O=gpurun_out/a1.txt
( python -m pytest tests/test_gpu_parity.py -x -q -s -k "ill_conditioned" 2>&1 | grep -v amdgpu.ids | tail -15 ) > $O
for e in "X=1" "GPC_LOOKAHEAD=1" "GPC_LOOKAHEAD=1 GPC_PANEL_FLOW_LEAN=1" "GPC_PANEL_FLOW_LEAN=1"; do
  echo "== $e" >> $O
  env $e python tools/factor_sweep.py 4096 6144 8192 12288 16384 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
