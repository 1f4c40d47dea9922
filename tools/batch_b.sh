cd $GRAFT_REPO_ROOT
O=gpurun_out/b14.txt; : > $O
for rep in 1 2; do
for t in "" "4096=4096,8192=1280,28672=1024,1000000=1536" "4096=4096,8192=1280,20480=1024,1000000=1536"; do
  echo -n "[$t]: " >> $O
  GPC_BENCH_PHASES=0 GPC_NB_TABLE="$t" python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['launches_per_step'], r['avg_launch_ms'], r['mfma_f64_probe_tflops'])" >> $O
done
done
for nb in 1024 1536 2048; do echo "LAUUM_NB=$nb" >> $O; GPC_POTRI_LAUUM_NB=$nb python tools/potri_bench.py 65536 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-85 >> $O; done
