# kernel + memory-copy trace of the GP-LVM run (config 5): the timeline of one evaluation cycle with its idle gaps (GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/kt_lvm
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kt_lvm -o t -- $R/gpc_amd/host/gplvm -v 3 -s 1 learn -k rbf -i 1 -# 100 $R/tests/golden/oilTrain.svml /tmp/oil.model > /tmp/lvm.out 2>&1
python - /tmp/kt_lvm <<PY > $R/gpurun_out/r55_lvm.txt
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(f); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
    print([t for t in tabs if 'rocpd' not in t or t.count('_') < 4][:40])
    try:
        rows = cur.execute("select start, end, name from kernels order by start").fetchall()
    except Exception as e:
        print("kernels view failed", e); rows = []
    try:
        mrows = cur.execute("select start, end, name from memory_copies order by start").fetchall()
    except Exception as e:
        print("memory_copies view failed", e); mrows = []
    ev = sorted([(s, e, n) for s, e, n in rows] + [(s, e, "MEMCPY " + str(n)) for s, e, n in mrows])
    n = len(ev)
    lo = n * 2 // 3
    t0 = ev[lo][0]
    prev = None
    for s, e, name in ev[lo: lo + 70]:
        gap = (s - prev) / 1e3 if prev else 0.0
        print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, name[:70]))
        prev = e
PY
