#!/bin/bash
# kernel-trace only (fast). usage: trace.sh TAG WORKLOAD STEPS [env...]
TAG=$1; WL=$2; STEPS=$3
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}
mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    print("== top kernels", f.split('/')[-1])
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 12"):
        print("%-70s calls %6d total_us %10.1f avg_us %9.2f  %5.1f%%" % (r[0][:70], r[1], r[2]/1.0, r[3], r[4]))
PY
