#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of tools/bench_configs.py -> gpurun_out/prof_configs_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_configs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_configs.py --steps 3 --warmup 1 > $OUT/trace.log 2>&1
find $OUT -name "*kernel_stats.csv" | head
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("# rocprofv3 --kernel-trace --stats of tools/bench_configs.py (all supplementary workloads in one process)")
    for r in rows[:25]:
        print("%-110s calls=%-5s avg_ns=%-14s total_ns=%-14s pct=%s" % (r["Name"][:110], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
PY
