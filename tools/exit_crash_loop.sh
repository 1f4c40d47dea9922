#!/bin/bash
# run `gp learn` on the sinc data N times and count the runs that do not exit with status 0 (a crash at process exit shows up here)
n=${1:-100}; bad=0
for i in $(seq 1 $n); do
  $GRAFT_REPO_ROOT/gpc_amd/host/gp -s 1 learn -# 30 $GRAFT_REPO_ROOT/tests/golden/sinc.svml /tmp/m.model > /tmp/gp_out.txt 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i: exit status $rc"; tail -2 /tmp/gp_out.txt; fi
done
echo "$bad of $n runs failed"
