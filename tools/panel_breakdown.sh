#!/bin/bash
# one cfg-3 factor under a kernel trace: what lies BETWEEN the trailing updates (per panel: kernels, copies, gaps)
export TMPDIR=/tmp GPC_BENCH_PHASES=0
rm -rf /tmp/tl_pb; cd /tmp
( cd $GRAFT_REPO_ROOT; rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_pb -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline ) > /tmp/tl_pb.out 2>&1
python - <<'PY'
import sqlite3, glob, collections
for f in glob.glob("/tmp/tl_pb/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    # the last factor: from the last gram_sym_kernel on
    g = max(i for i, r in enumerate(rows) if "gram_sym" in r[0])
    rows = rows[g:]
    upd = [i for i, r in enumerate(rows) if "gemm_nt_ring_kernel<1>" in r[0] or "gemm_nt_fast_kernel<4, 1" in r[0]]
    tot = collections.Counter(); cnt = collections.Counter(); gaps = 0.0
    prev_end = rows[0][2]
    seg = []
    for a, b in zip([0] + upd, upd + [len(rows)]):
        pass
    last = rows[0][2]
    panel_ms = []
    k = 0
    for i, (n, s, e) in enumerate(rows[1:], 1):
        if i in upd:
            last = e
            continue
        key = n.split("(")[0][-60:]
        tot[key] += (e - s) / 1e6; cnt[key] += 1
        gaps += max(0.0, (s - last) / 1e6)
        last = max(last, e)
    span = (rows[-1][2] - rows[0][1]) / 1e6
    up = sum((rows[i][2] - rows[i][1]) / 1e6 for i in upd)
    print("factor span %.1f ms, trailing updates %.1f ms in %d launches, everything else %.1f ms" % (span, up, len(upd), span - up))
    for key, v in tot.most_common(14):
        print("  %8.2f ms  %4d x  %s" % (v, cnt[key], key))
    print("  idle gaps between non-update kernels (host / copies not in the kernel table): %.2f ms" % gaps)
    # per tall panel: time from the end of an update to the start of the next
    ends = [rows[i][2] for i in upd]; starts = [rows[i][1] for i in upd]
    between = [(starts[i + 1] - ends[i]) / 1e6 for i in range(len(upd) - 1)]
    print("  between consecutive updates (ms):", " ".join("%.2f" % b for b in between))
PY
