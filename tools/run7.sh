cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r7_bench_default.json 2> gpurun_out/r7_bench_default.err
for pad in 0 16 264 2064; do GPC_BENCH_LDPAD=$pad GPC_BENCH_PHASES=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ldpad $pad: ms/step %.1f  gram %.3f ms (%.0f GB/s)  syrk %.2f TF' % (l['ms_per_step'], l['roofline']['gram']['avg_launch_ms'], l['roofline']['gram']['achieved'], l['roofline']['achieved']))
"; done > gpurun_out/r7_ldpad.txt 2>&1
for dbg in 0 2 3; do GPC_GRAM_DEBUG=$dbg python tools/gram_bench.py 65536 32; GPC_GRAM_DEBUG=$dbg python tools/gram_bench.py 65536 8; done > gpurun_out/r7_gram_dbg.txt 2>&1
