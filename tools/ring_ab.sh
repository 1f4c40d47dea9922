#!/bin/bash
# A/B of the trailing update's kernel: 128 x 128 tiles, two workgroups per CU, registers two stages ahead (default) against the
# ring form (GPC_GEMM_RING=1: 256 x 128 per CU, sixteen waves, global_load_lds into three LDS stages, persistent)
GPC_GEMM_RING=1 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm or syrk" 2>&1 | tail -2
for v in "0 1" "1 0" "1 1"; do
  set -- $v
  echo "== GPC_GEMM_RING=$1 STAGGER=$2"
  GPC_GEMM_RING=$1 GPC_GEMM_RING_STAGGER=$2 python tools/syrk_k_sweep.py ${M:-32768} 2>&1 | grep "beta=1" | grep -v "K= 128\|K= 256"
done
