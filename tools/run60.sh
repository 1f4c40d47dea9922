cd $GRAFT_REPO_ROOT
for G in 1 0; do
echo "== GATHER=$G" >> gpurun_out/r60.txt
for poison in 0 1 1 0; do
  GPC_HOST_GATHER=$G GPC_POISON_ALLOC=$poison gpc_amd/host/gp -s 1 learn -# 30 tests/golden/sinc.svml /tmp/m$poison.model > /tmp/out$poison.txt 2>&1
  echo "poison=$poison rc=$? $(grep -v '^#' /tmp/m$poison.model | md5sum) $(tail -1 /tmp/out$poison.txt)" >> gpurun_out/r60.txt
done
done
