#!/bin/bash
# usage (on the GPU box): bash tools/trace.sh <label> <command...>   -> prints the top kernels of the command
export TMPDIR=/tmp
L=$1; shift
rm -rf /tmp/tr_$L; cd /tmp
( cd ${GRAFT_REPO_ROOT:-.}; rocprofv3 --kernel-trace --stats -d /tmp/tr_$L -o t -- "$@" ) > /tmp/tr_$L.out 2>&1
python - /tmp/tr_$L <<'PY'
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    print("%-80s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 12"):
        print("%-80s %8d %12.2f %10.1f %6.1f%%" % (r[0][:80], r[1], r[2] / 1e3, r[3], r[4]))
PY
