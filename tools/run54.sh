export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/kt_lvm
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_lvm -o t -- $R/gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 $R/tests/golden/oilTrain.svml /tmp/oil.model > /tmp/lvm.out 2>&1
grep -i "evaluations" /tmp/lvm.out > $R/gpurun_out/r54_lvm.txt
python - /tmp/kt_lvm <<PY >> $R/gpurun_out/r54_lvm.txt
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(f); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t.lower()]
    print("tables:", kd[:6])
    print("%-90s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 25"):
        print("%-90s %8d %12.3f %10.2f %6.1f%%" % (r[0][:90], r[1], r[2] / 1e6, r[3] / 1e3, r[4]))
    # timeline
    for t in kd:
        try:
            cols = [c[1] for c in cur.execute("pragma table_info('%s')" % t)]
            if 'start' in cols and 'end' in cols:
                rows = sorted(cur.execute("select start, end from %s" % t).fetchall())
                # the last 60% of the run (optimisation loop)
                n = len(rows); rows2 = rows[n // 3:]
                busy = sum(e - s for s, e in rows2); span = rows2[-1][1] - rows2[0][0]
                gaps = [rows2[i + 1][0] - rows2[i][1] for i in range(len(rows2) - 1)]
                import statistics
                print("table %s: %d dispatches (last two thirds: %d): busy %.3f ms of span %.3f ms (%.1f%%), median gap %.2f us, mean gap %.2f us" % (t, n, len(rows2), busy / 1e6, span / 1e6, 100.0 * busy / span, statistics.median(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3))
                break
        except Exception as e:
            print("err", t, e)
PY
