cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 1" "1 0" "0 1"; do
  set -- $cfg
  echo "GPC_LOOKAHEAD=$1 GPC_PANEL_FLOW_LEAN=$2" >> gpurun_out/r40.txt
  GPC_LOOKAHEAD=$1 GPC_PANEL_FLOW_LEAN=$2 GPC_BENCH_PHASES=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('whole_factor_frac_of_n_gpu_peak'), d['roofline'].get('avg_launch_ms'), d['roofline'].get('launches_per_step'))" >> gpurun_out/r40.txt
done
