#!/bin/bash
# A/B of the 128 x 128 kernel (GPC_GEMM_RING=0 keeps every product on it): this build against libgpc_hip_prev.so
python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or syrk or potrf or chol or potri or trsm" 2>&1 | tail -2
for i in 1 2; do for v in "" prev; do echo "[$v]"; GPC_GEMM_RING=0 GPC_LIB_VARIANT=$v python tools/ring_msweep.py 2>&1 | grep "M= 8192\|M=16384\|M=32768"; done; done
for v in "" prev; do echo "[$v]"; GPC_LIB_VARIANT=$v python tools/gemm_forms.py 2>/dev/null | tail -6; GPC_LIB_VARIANT=$v python bench.py --workload cfg2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', j['ms_per_step'], j['phases']['potri_ms'])"; done
