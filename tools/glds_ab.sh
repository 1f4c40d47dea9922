#!/bin/bash
# A/B of the trailing update's staging: registers two stages ahead (default) against global_load_lds one stage ahead (GPC_GEMM_GLDS=1)
for v in 0 1; do
  echo "== GPC_GEMM_GLDS=$v"
  GPC_GEMM_GLDS=$v python tools/syrk_k_sweep.py 32768 2>&1 | grep "beta=1"
done
GPC_GEMM_GLDS=1 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or syrk or potrf" 2>&1 | tail -3
for v in 0 1; do
  echo "== bench GPC_GEMM_GLDS=$v"
  GPC_GEMM_GLDS=$v GPC_BENCH_PHASES=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['frac'], j['roofline']['mfma_f64_probe_tflops'])"
done
