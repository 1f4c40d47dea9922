#!/bin/bash
# usage (on the GPU box): bash tools/trace_list.sh <label> <n> <command...>  -> the last n kernel dispatches of the command, in order, with durations
export TMPDIR=/tmp
L=$1; NLAST=$2; shift 2
rm -rf /tmp/tl_$L; cd /tmp
( cd ${GRAFT_REPO_ROOT:-.}; rocprofv3 --kernel-trace -d /tmp/tl_$L -o t -- "$@" ) > /tmp/tl_$L.out 2>&1
python - /tmp/tl_$L $NLAST <<'PY'
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % v[0])]
    rows = cur.execute("select name, start, end, grid_x from %s order by start" % v[0]).fetchall() if "grid_x" in cols else cur.execute("select name, start, end, 0 from %s order by start" % v[0]).fetchall()
    t0 = rows[-int(sys.argv[2])][1]
    for n, s, e, g in rows[-int(sys.argv[2]):]:
        print("%9.1f us  +%8.1f us  grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, g, n[:90]))
PY
