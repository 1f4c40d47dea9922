#!/bin/bash
# Timing-only ablations of the kernel-parameter gradient kernels at N = 65 536, D = 32 (variants built by tools/build_variant.sh with
# -DGPC_KG_ABL_NOEXP / _NOLOAD / _NOMMA / _NOY; their results are wrong by construction).  usage (GPU box): bash tools/kgrad_abl.sh
for v in "" kgnoexp kgnoload kgnomma kgnoy kgnone; do
  [ -n "$v" ] && [ ! -f gpc_amd/lib/libgpc_hip_$v.so ] && continue
  echo "== variant '${v}'"
  GPC_LIB_VARIANT=$v python tools/grad_bench.py 65536 ${D:-32} 2>&1 | grep "kern_grad "
done
