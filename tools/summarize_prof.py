"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print("# rocprofv3 summary of", os.path.basename(root))
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("\n## kernel stats (%s)" % os.path.relpath(f, root))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print("  %-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"),
                                                            r.get("AverageNs"), r.get("Percentage")))
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    seen = set()
    print("\n## per-dispatch resources (%s)" % os.path.relpath(f, root))
    for r in rows:
        k = r.get("Kernel_Name", "")
        if k in seen:
            continue
        seen.add(k)
        print("  %-60s grid=%s wg=%s VGPR=%s accVGPR=%s SGPR=%s LDS=%s scratch=%s" % (
            k[:60], r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
            r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r.get("Kernel_Name", "")][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0)))
        print("\n## counters (%s)" % os.path.relpath(f, root))
        for k, cs in agg.items():
            if not any(x in k for x in ("pbs_kernel", "keyswitch", "external_product", "unfold2")):
                continue
            for c, v in cs.items():
                print("  %-50s %-28s per-dispatch mean=%.6g  (n=%d)" % (k[:50], c, sum(v) / len(v), len(v)))
