#!/bin/bash
# Produces the rocprofv3 summaries committed under profiles/ (run on the GPU box: gpurun -- bash tools/make_profiles.sh r01)
TAG=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT; cd /tmp
summ() { # db-dir label
python - "$1" "$2" <<PY
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    print("# %s : rocprofv3 --kernel-trace --stats (top kernels; durations in ms)" % sys.argv[2])
    print("%-72s %8s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 14"):
        print("%-72s %8d %14.1f %12.2f %6.1f%%" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
}
# 1. kernel traces of the bench command itself
for wl in cfg3 cfg2; do
  steps=1; [ $wl = cfg2 ] && steps=3
  rm -rf /tmp/kt_$wl
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -o t -- python $R/bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline > $OUT/bench_${wl}_under_trace.json 2> /dev/null
  summ /tmp/kt_$wl "python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline" > $OUT/kernel_trace_${wl}.txt
done
# 2. PMC passes on ONE big trailing-update launch (tools/one_syrk.py: M=32768, K=1536 -- the ring kernel -- 3 launches), separate runs
pmc() { # name counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --pmc "$@" -d /tmp/pmc_$n -o p -- python $R/tools/one_syrk.py 2 32768 1536 > /dev/null 2>&1
  echo "# rocprofv3 --pmc $* -- python tools/one_syrk.py 2 32768 1536   (gemm_nt_ring_kernel; per-dispatch sums over all XCDs/SEs)" > $OUT/pmc_syrk_$n.txt
  python $R/tools/pmc_query.py /tmp/pmc_$n gemm >> $OUT/pmc_syrk_$n.txt 2>&1
}
pmc mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum
# 3. the Gram kernel alone (HBM bytes)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_g$c
  timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_g$c -o p -- python $R/tools/gram_bench.py 65536 32 > /dev/null 2>&1
  echo "# rocprofv3 --pmc $c -- python tools/gram_bench.py 65536 32   (cfg 3's Gram: gram_sym_kernel; counter values in KB summed over the XCDs)" > $OUT/pmc_gram_$c.txt
  python $R/tools/pmc_query.py /tmp/pmc_g$c gram_sym >> $OUT/pmc_gram_$c.txt 2>&1
done
# 4. dpotri alone at cfg 3's size (in place from N = 24 576): kernel trace
rm -rf /tmp/kt_potri
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_potri -o t -- python $R/tools/potri_only.py 65536 > /dev/null 2>&1
summ /tmp/kt_potri "python tools/potri_only.py 65536   (factor once, then 4 x [copy + gpc_potri_f64 in place])" > $OUT/kernel_trace_potri.txt
# 5. PMC passes: the dL/dX walk (pair_walk_kernel, N = 32 768, D = 8 and 16), the dataflow panel kernel on a 1024 x 1024 tile
pmcx() { # file pattern counters -- command...
  f=$1; pat=$2; set=$3; shift 3
  rm -rf /tmp/pmcx
  timeout 300 rocprofv3 --pmc $set -d /tmp/pmcx -o p -- "$@" > /dev/null 2>&1
  echo "# --pmc $set" >> $f
  python $R/tools/pmc_query.py /tmp/pmcx "$pat" >> $f 2>&1
}
F=$OUT/pmc_gradx.txt
echo "# rocprofv3 --pmc <counters> -- python tools/grad_bench.py 32768 <D>   (pair_walk_kernel: gpc_kern_gradx_f64; per-dispatch sums over all XCDs; FETCH_SIZE / WRITE_SIZE in KB)" > $F
for D in 8 16; do
  echo "# ---- D = $D" >> $F
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64" "FETCH_SIZE" "WRITE_SIZE"; do
    pmcx $F pair_walk "$set" python $R/tools/grad_bench.py 32768 $D
  done
done
F=$OUT/pmc_panel_flow.txt
echo "# rocprofv3 --pmc <counters> -- python tools/flow_check.py 1024 child   (panel_flow_kernel on a 1024 x 1024 factorisation: 136 workgroups, 16 dependent steps)" > $F
for set in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM"; do
  pmcx $F panel_flow "$set" python $R/tools/flow_check.py 1024 child
done
ls -la $OUT
