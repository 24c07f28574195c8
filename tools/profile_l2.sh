#!/bin/bash
# L2 hit / miss of the bootstrap kernel for one batch size: tools/profile_l2.sh <B> <set>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_l2_$1_$2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/tools/gpu_perf.py $1 $2 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pbs_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
h, m = (sum(agg[k]) / len(agg[k]) for k in ("TCC_HIT_sum", "TCC_MISS_sum"))
print("hit %.3g miss %.3g hit-rate %.1f%%  miss bytes %.1f GB" % (h, m, 100 * h / (h + m), m * 128 / 1e9))
PY
