#!/bin/bash
# dpotri in place at N = 65 536 under other panel widths of its two phases (GPC_TRTRI_NB: V = L^-T; GPC_POTRI_LAUUM_NB: V V').
# usage (GPU box): bash tools/potri_nb_sweep.sh
for cfg in "0 0" "1024 1536" "1536 1536" "1024 2048" "1536 2048" "2048 2048"; do
  set -- $cfg
  echo "== GPC_TRTRI_NB=$1 GPC_POTRI_LAUUM_NB=$2"
  if [ "$1" = 0 ]; then python tools/potri_time.py ${N:-65536}; else GPC_TRTRI_NB=$1 GPC_POTRI_LAUUM_NB=$2 python tools/potri_time.py ${N:-65536}; fi
done
