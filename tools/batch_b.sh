cd $GRAFT_REPO_ROOT
O=gpurun_out/b19.txt; : > $O
for v in "" xnobar xst12 xst24 xst40 xst64; do
  echo "== [$v]" >> $O
  for D in 16 32; do GPC_LIB_VARIANT=$v python tools/grad_bench.py 65536 $D 2>/dev/null | grep "kern_grad rbfard" >> $O; done
done
