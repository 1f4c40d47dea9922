#!/bin/bash
# PMC passes on the rbfard parameter-gradient kernel at cfg 3's size (N = 65 536, D = 32), for the four-wave form (one wave per
# SIMD) and the eight-wave form (two): waves / busy / wait cycles, MFMA busy, instruction mix, LDS, HBM bytes.  Separate
# rocprofv3 runs per counter group.  usage (GPU box): bash tools/pmc_kgrad.sh r03   -> gpurun_out/profiles_r03/pmc_kgrad_ard.txt
TAG=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT; cd /tmp
F=$OUT/pmc_kgrad_ard.txt
echo "# rocprofv3 --pmc <counters> -- python tools/ard_grad_one.py 65536 32 2   (kern_grad_ard_sym_kernel; per-dispatch sums over all XCDs / SEs)" > $F
for nw in 0; do
  echo "# ---- default form at D = 32: eight waves of 32 x 32 patches, one workgroup per CU; MODE 1 = interior tiles (250 registers), MODE 2 = diagonal blocks" >> $F
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); rm -rf /tmp/pk_$i
    GPC_KGRAD_ARD_NW=$nw timeout 300 rocprofv3 --pmc $set -d /tmp/pk_$i -o p -- python $R/tools/ard_grad_one.py 65536 32 2 > /dev/null 2>&1
    echo "# --pmc $set" >> $F
    python $R/tools/pmc_query.py /tmp/pk_$i kern_grad_ard >> $F 2>&1
  done
done
cat $F
