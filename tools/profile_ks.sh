#!/bin/bash
# Runs on the GPU box (via gpurun): kernel stats + LDS / VALU / memory counters of the circuit-bootstrap workload (packing key switch)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/gpu_perf_cb.py 1024"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "keyswitch|kernel stats|counters" $OUT/summary.txt
