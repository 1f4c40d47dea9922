cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r18_bench_default.json 2> gpurun_out/r18_bench_default.err
python bench.py --workload cfg2 --steps 20 --warmup 3 > gpurun_out/r18_bench_cfg2.json 2>/dev/null
