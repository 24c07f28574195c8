#!/bin/bash
# Runs on the GPU box (via gpurun): counters of the N = 2048 bootstrap kernel (tools/gpu_perf.py 4096 lvl2)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_lvl2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/gpu_perf.py 4096 lvl2"
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "pbs_kernel" $OUT/summary.txt
