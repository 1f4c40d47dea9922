#!/bin/bash
# usage (on the GPU box): bash tools/trace_span.sh <label> <command...>  -> dispatch count, busy time, first-to-last span
export TMPDIR=/tmp
L=$1; shift
rm -rf /tmp/ts_$L; cd /tmp
rocprofv3 --kernel-trace -d /tmp/ts_$L -o t -- "$@" > /tmp/ts_$L.out 2>&1
python - /tmp/ts_$L <<'PY'
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tabs if x.startswith("kernels") or x == "kernels"]
    name = "kernels" if "kernels" in tabs else (t[0] if t else None)
    if not name:
        print("tables:", tabs[:40]); continue
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % name)]
    rows = list(cur.execute("select start, end from %s order by start" % name))
    busy = sum(e - s for s, e in rows)
    # union of busy intervals (kernels of two streams may overlap)
    un, cs, ce = 0, None, None
    for s, e in rows:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: un += ce - cs; cs, ce = s, e
    if cs is not None: un += ce - cs
    gaps = sorted(((rows[i + 1][0] - max(r[1] for r in rows[:i + 1][-4:])) for i in range(len(rows) - 1)), reverse=True)
    print("dispatches %d  sum of durations %.3f ms  union busy %.3f ms  first-to-last %.3f ms" % (len(rows), busy / 1e6, un / 1e6, (rows[-1][1] - rows[0][0]) / 1e6))
    print("largest idle gaps (us):", [round(g / 1e3, 1) for g in gaps[:12]])
PY
