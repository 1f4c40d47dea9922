#!/bin/bash
# factor time (tools/flow_check.py child) under combinations of panel width and look-ahead: flow_sweep.sh "<nb list>" N...
nbs=$1; shift
for n in "$@"; do
  for nb in $nbs; do
    echo -n "N=$n NB=$nb: "
    GPC_LOOKAHEAD=${LA:-0} GPC_NB=$nb GPC_PANEL_FLOW=1 timeout 200 python tools/flow_check.py $n child /tmp/x.npy | tail -1
  done
done
