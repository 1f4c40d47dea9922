#!/bin/bash
# Looks for the exit-time crash of DESIGN.md section 5c: runs `gp learn` on the sinc data N times while ANOTHER process holds the
# GPU (as the test harness does), leaving through the ordinary exit path (GPC_EXIT=return: atexit handlers -- gpc_shutdown --
# and static destructors) or through _exit (default), and counts the runs that do not end with status 0.  With the
# LD_PRELOAD aid tools/segv/libsegv_bt.so a crashing run leaves its backtrace in gpurun_out/exit_bt.txt.
# usage: tools/exit_crash_loop.sh N [return|fast] [holder: 1|0]
n=${1:-100}; mode=${2:-return}; holder=${3:-1}; bad=0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
if [ "$holder" = "1" ]; then
  python -c "import torch,time; x=torch.zeros(1<<20,device='cuda'); [ (x.add_(1), torch.cuda.synchronize(), time.sleep(0.01)) for _ in range(10**7) ]" &
  hp=$!
  sleep 8
fi
if [ -f $R/tools/segv/libsegv_bt.so ]; then pre=$R/tools/segv/libsegv_bt.so; fi
for i in $(seq 1 $n); do
  if [ "$mode" = "return" ]; then
    SEGV_BT_FILE=$R/gpurun_out/exit_bt.txt LD_PRELOAD=$pre GPC_EXIT=return $R/gpc_amd/host/gp -s 1 learn -# 30 $R/tests/golden/sinc.svml /tmp/m.model > /tmp/gp_out.txt 2>&1
  else
    $R/gpc_amd/host/gp -s 1 learn -# 30 $R/tests/golden/sinc.svml /tmp/m.model > /tmp/gp_out.txt 2>&1
  fi
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i: exit status $rc"; tail -2 /tmp/gp_out.txt; fi
done
[ -n "$hp" ] && kill $hp
echo "mode=$mode holder=$holder: $bad of $n runs failed"
