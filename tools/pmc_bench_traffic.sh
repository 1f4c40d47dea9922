#!/bin/bash
# HBM traffic of the dominant kernel in the bench command itself: two separate --pmc passes (FETCH_SIZE costs 3 of the
# 4 TCC slots, WRITE_SIZE 2), no tracing domain besides the counters.  Run on the GPU box:
#   gpurun -- bash tools/pmc_bench_traffic.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT; cd /tmp
# GPC_BENCH_PHASES=0: nothing but the timed steps runs, so the dispatches named gemm_nt_ring_kernel<1, are exactly the trailing
# updates of ONE factor that take the ring kernel (round 5: every update with >= 5120 tiles of 256 x 128 = 18 432 rows; the smaller ones stay on
# gemm_nt_fast_kernel<4, 1, ...> and are collected beside them)
export GPC_BENCH_PHASES=0
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  timeout 420 rocprofv3 --pmc $c -d /tmp/pmcb_$c -o p -- $CMD > $OUT/pmc_bench_${c}_stdout.json 2> /dev/null
  echo "rc=$? for $c"
done
python - $OUT/pmc_bench_traffic.json <<'PY'
import sqlite3, glob, json, sys, collections
res = {"command": "GPC_BENCH_PHASES=0 rocprofv3 --pmc <C> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline  (C = FETCH_SIZE, WRITE_SIZE; separate passes)",
       "kernel": "gemm_nt_ring_kernel<1, false> (the trailing-update launches that take the ring kernel)", "kernel_pattern": "gemm_nt_ring_kernel<1,",
       "units": "counter values are KB summed over the 8 XCDs per dispatch"}
def collect(pattern):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n, dur = 0.0, 0, 0.0
        for f in glob.glob("/tmp/pmcb_%s/**/*.db" % c, recursive=True):
            cur = sqlite3.connect(f).cursor()
            byd = collections.defaultdict(lambda: [0.0, 0.0, ""])
            for did, kn, cn, v, d in cur.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
                if cn == c and pattern in kn:
                    byd[did][0] += v; byd[did][1] = d
            for did, (v, d, _) in byd.items():
                tot += v; n += 1; dur += d
        out[c] = {"dispatches": n, "sum_kb": tot, "avg_kb_per_dispatch": tot / max(n, 1), "sum_duration_ms_under_pmc": dur / 1e6}
    return out
res.update(collect("gemm_nt_ring_kernel<1,"))
small = collect("gemm_nt_fast_kernel<4, 1,")
res["smaller_updates_gemm_nt_fast_kernel"] = dict(small, hbm_bytes_per_launch=(2.0 * small["FETCH_SIZE"]["avg_kb_per_dispatch"] + small["WRITE_SIZE"]["avg_kb_per_dispatch"]) * 1024.0)
f, w = res["FETCH_SIZE"], res["WRITE_SIZE"]
# MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> double it; WRITE_SIZE taken as is
res["hbm_bytes_per_launch"] = (2.0 * f["avg_kb_per_dispatch"] + w["avg_kb_per_dispatch"]) * 1024.0
res["correction"] = "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"
# the O(N^2) kernel of the same command: the mirrored Gram build (algorithmic bytes 8 N^2 + 8 N D = 34.38 GB at cfg 3)
g = collect("gram_sym_kernel")
res["gram_sym_kernel"] = dict(g, hbm_bytes_per_launch=(2.0 * g["FETCH_SIZE"]["avg_kb_per_dispatch"] + g["WRITE_SIZE"]["avg_kb_per_dispatch"]) * 1024.0,
                              algorithmic_bytes=8.0 * 65536.0 ** 2 + 8.0 * 65536.0 * 32.0)
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res))
PY
