cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 12 16; do python tools/grad_bench.py 32768 $d 2>&1 | grep gradx; GPC_PAIR_WALK_FORM=1 python tools/grad_bench.py 32768 $d 2>&1 | grep gradx | sed 's/$/   [form 1]/'; done > gpurun_out/r5_gradx.txt
bash tools/trace.sh dtc python tools/dtc_bench.py > gpurun_out/r5_trace_dtc.txt 2>&1
