#!/bin/bash
# Runs on the GPU box (via gpurun): round-6 rocprofv3 summaries of every kernel DESIGN.md section 4 names, from the CURRENT tree.
#   tools/profile_r06.sh <tag> ["pbs ep ep2 lvl2 ks cb unf"]  ->  gpurun_out/prof_<tag>_<name>/summary.txt  (copy into profiles/)
# Kernel-trace statistics and counters are taken in SEPARATE runs; every --pmc pass is its own run (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <name> <counter sets separated by ;> <program args...>
  local NAME=$1 SETS=$2; shift 2
  local OUT=$ROOT/gpurun_out/prof_${TAG}_$NAME
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1
  IFS=';' read -ra LIST <<< "$SETS"
  for C in "${LIST[@]}"; do
    local N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 "$@" > $OUT/pmc_$N.log 2>&1
  done
  python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
}
ONLY=${2:-"pbs ep ep2 lvl2 cb split"}
want() { case " $ONLY " in *" $1 "*) return 0;; esac; return 1; }
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"
LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"
ALL="FETCH_SIZE;WRITE_SIZE;$LDS;$SQ1;TCC_HIT_sum TCC_MISS_sum;$SQ2"
want pbs && prof pbs "$ALL" $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1
want ep && EP_GROUPS=60 prof ep "FETCH_SIZE;WRITE_SIZE;$LDS;$SQ2" $ROOT/tools/gpu_perf_ep.py 65536 set1
want ep2 && EP_GROUPS=60 prof ep2 "FETCH_SIZE;WRITE_SIZE;$SQ1;$SQ2" $ROOT/tools/gpu_perf_ep.py 16384 lvl2
want lvl2 && prof lvl2 "$ALL" $ROOT/tools/gpu_perf.py 4096 lvl2
want ks && prof ks "FETCH_SIZE;$LDS;$SQ2" $ROOT/tools/gpu_perf_ks.py 4096 lvl2 device
want cb && prof cb "FETCH_SIZE;$SQ1;$SQ2" $ROOT/tools/gpu_perf_cb.py 1024
want split && prof split "FETCH_SIZE;$SQ1;TCC_HIT_sum TCC_MISS_sum;$SQ2" $ROOT/tools/gpu_perf_split.py
want unf && prof unf "FETCH_SIZE;$SQ1;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;$SQ2" $ROOT/tools/gpu_perf_unfold.py 4096 lvl2
for n in $ONLY; do echo "=== $n"; grep -E "kernel stats|calls=|per-dispatch" $ROOT/gpurun_out/prof_${TAG}_$n/summary.txt | head -40; done
