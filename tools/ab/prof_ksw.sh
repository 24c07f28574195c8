#!/bin/bash
# on the GPU box: per-kernel times (rocprofv3 kernel trace) of tools/ab/_build/ksw_<variant> for one case;  tools/ab/prof_ksw.sh "<variants>" <case args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=$1; shift
mkdir -p $ROOT/gpurun_out/r06b
exec > >(tee $ROOT/gpurun_out/r06b/${KSW_PROF_OUT:-ksw_prof.txt}) 2>&1
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  rm -rf /tmp/prof_$v
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- $ROOT/tools/ab/_build/ksw_$v "$@" > /tmp/prof_$v.log 2>&1
  echo "== $v  ($*)"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])): print('  %-70s calls=%4s avg_us=%10.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))" "$f"; else echo "(no stats file)"; tail -5 /tmp/prof_$v.log; fi
done
