cd $GRAFT_REPO_ROOT
for I in "" 0 1; do
  echo "== GPC_FLOW_GIVEN_INV='$I'" >> gpurun_out/r51.txt
  for N in 2048 4096 6144 8192 16384; do
    GPC_FLOW_GIVEN_INV=$I python tools/potri_bench.py $N 2>/dev/null | tail -1 >> gpurun_out/r51.txt
  done
  GPC_FLOW_GIVEN_INV=$I python tools/posterior_bench.py 2>/dev/null | tail -3 >> gpurun_out/r51.txt
done
