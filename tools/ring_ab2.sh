#!/bin/bash
for v in 0 1; do echo "== GPC_GEMM_RING=$v"; GPC_GEMM_RING=$v python tools/ring_msweep.py; done
