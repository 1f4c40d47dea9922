cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for D in 32 8; do for dbg in 0 2 4 3 5 6; do GPC_GRAM_DEBUG=$dbg python tools/gram_bench.py 65536 $D 2>/dev/null; done; done > gpurun_out/r8_gram_dbg.txt 2>&1
python tools/fill_bw.py >> gpurun_out/r8_gram_dbg.txt 2>&1
