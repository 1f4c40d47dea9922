cd $GRAFT_REPO_ROOT
run() { # label envs...
  label=$1; shift
  for i in $(seq 1 150); do
    env "$@" GPC_POISON_ALLOC=$((i % 2)) gpc_amd/host/gp -s 1 learn -# 30 tests/golden/sinc.svml /tmp/m.model > /dev/null 2>&1
    grep -v '^#' /tmp/m.model | md5sum
  done | sort | uniq -c | sed "s/^/$label /" >> gpurun_out/r62.txt
}
run default GPC_DUMMY=1
run gather0 GPC_HOST_GATHER=0
run flow0 GPC_PANEL_FLOW=0
