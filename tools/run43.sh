export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/kt_p8
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_p8 -o t -- python $R/tools/potri_only.py 8192 > /dev/null 2>&1
python - /tmp/kt_p8 <<PY > $R/gpurun_out/r43_potri8192.txt
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    print("%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 20"):
        print("%-100s %8d %14.3f %12.3f %6.1f%%" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
