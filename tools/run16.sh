cd $GRAFT_REPO_ROOT
python bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r16_bench_cfg4.json 2> gpurun_out/r16_bench_cfg4.err
python bench.py --workload cfg2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r16_bench_cfg2.json 2> gpurun_out/r16_bench_cfg2.err
