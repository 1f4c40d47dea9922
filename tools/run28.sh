cd $GRAFT_REPO_ROOT
for v in "" abl1 abl2 abl3 abl4; do echo -n "variant=[$v] "; GPC_LIB_VARIANT=$v python tools/grad_bench.py 65536 32 2>/dev/null | grep "kern_grad rbfard"; done > gpurun_out/r28_ard_abl.txt 2>&1
