cd $GRAFT_REPO_ROOT
O=gpurun_out/b2.txt; : > $O
for mr in 1000000 3072 4096 4608 5120; do
  echo "== LEAN_MINROWS=$mr" >> $O
  GPC_PANEL_FLOW_LEAN_MINROWS=$mr python tools/factor_sweep.py 1000 2048 3072 4096 4608 5120 6144 8192 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
for mr in 1000000 1 2048 3072 4096; do
  echo "== potri GIVEN_LEAN_MINROWS=$mr" >> $O
  for n in 1000 2048 3072 4096 6144 8192; do GPC_PANEL_FLOW_LEAN_MINROWS=5120 GPC_FLOW_GIVEN_LEAN_MINROWS=$mr python tools/potri_bench.py $n 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-75 >> $O; done
done
