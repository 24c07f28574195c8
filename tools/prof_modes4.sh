#!/bin/bash
# tools/prof_modes4.sh: per-launch L2 counters of tools/gpu_perf_modes4.py (8 lvl2 bootstraps back to back, then 8 with an extract kernel in between)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04_modes4b
for C in "TCC_PROBE_sum TCC_ALL_TC_OP_INV_EVICT_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum" "TCC_NC_REQ_sum TCC_RW_REQ_sum TCC_CC_REQ_sum TCC_UC_REQ_sum" "TCC_STREAMING_REQ_sum TCC_TAG_STALL_sum TCC_READ_sum TCC_EA0_RDREQ_128B_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum"; do
  N=$(echo $C | tr " " "_" | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -- python3 $GRAFT_REPO_ROOT/tools/gpu_perf_modes4.py > $OUT.$N.log 2>&1
done
ls $OUT
