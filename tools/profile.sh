#!/bin/bash
# tools/profile.sh -- rocprofv3 runs whose summaries are copied into profiles/ (run on the GPU box via gpurun).
#   pass 1: --kernel-trace --stats          (per-kernel time)
#   pass 2+: --pmc ...                      (counters; separate runs, never combined with tracing domains)
set -u
TAG=${1:-r01}
WL=${2:-cfg2}
STEPS=${3:-2}
EXTRA=${4:-}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace_err.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1_bench.json 2> $OUT/pmc1_err.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2_bench.json 2> $OUT/pmc2_err.log
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3_bench.json 2> $OUT/pmc3_err.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4_bench.json 2> $OUT/pmc4_err.log
find $OUT -name "*.csv" | head -30
ls -la $OUT/trace/* | head
