#!/bin/bash
# Timing-only ablations of the ring kernel's tile boundary (variants built by tools/build_variant.sh with -DGPC_RING_ABL_*; their
# results are wrong by construction).  usage (GPU box): bash tools/ring_abl.sh
for v in "" latewait store noepi; do
  echo "== variant '${v}'"
  GPC_LIB_VARIANT=$v python tools/ring_msweep.py 2>&1 | grep "M=32768\|M=49152\|M=16384"
done
