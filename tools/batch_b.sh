cd $GRAFT_REPO_ROOT
O=gpurun_out/b6.txt; : > $O
for mr in 12288 28672 1000000; do
  echo "== PANEL_INV_MINROWS=$mr" >> $O
  for n in 16384 24576 32768 49152 65536; do GPC_PANEL_INV_MINROWS=$mr python tools/potri_bench.py $n 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-85 >> $O; done
done
