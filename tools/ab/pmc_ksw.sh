#!/bin/bash
# on the GPU box: counters of table_ks_words_kernel (one rocprofv3 --pmc run per counter set);  tools/ab/pmc_ksw.sh <variant> <case args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; shift
mkdir -p $ROOT/gpurun_out/r06b
exec > >(tee $ROOT/gpurun_out/r06b/${KSW_PROF_OUT:-ksw_pmc.txt}) 2>&1
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rm -rf /tmp/pmc_$N
  timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$N -- $ROOT/tools/ab/_build/ksw_$V "$@" > /tmp/pmc_$N.log 2>&1
  f=$(find /tmp/pmc_$N -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 -c "
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'table_ks_words' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()): print('  %-28s per-dispatch mean = %.4g  (n=%d)' % (k, sum(v) / len(v), len(v)))" "$f"; else echo "(no counters for $C)"; tail -3 /tmp/pmc_$N.log; fi
done
