"""Turn a tools/profile_bench.sh summary into profiles/<tag>_traffic.json, the per-launch memory-side traffic of
the dominant kernel that bench.py reports as roofline.traffic.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB, taken in
separate --pmc passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly half the bytes of a
wide coalesced stream, so the read side is doubled; WRITE_SIZE is exact.  These are the L2's fabric-side
counters: Infinity-Cache hits are included, so this is an upper bound on true HBM traffic.
"""
import json
import re
import sys

summary, out, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
units = int(sys.argv[4]) if len(sys.argv) > 4 else None   # ciphertexts (units) per launch of the profiled run
vals = {}
for line in open(summary):
    m = re.search(r"(\S+)\s+per-dispatch mean=([0-9.e+]+)", line)
    if m and kernel in line:
        vals[m.group(1)] = float(m.group(2))
import os
stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mosfhet_amd", "libmosfhet_hip.so.srchash")
srchash = open(stamp).read().strip() if os.path.exists(stamp) else None   # bench.py reports `traffic` only for the build it was measured on
fetch = 2.0 * vals["FETCH_SIZE"] * 1024
write = vals["WRITE_SIZE"] * 1024
json.dump({"kernel": kernel, "source": summary, "srchash": srchash, **({"units_per_launch": units} if units else {}), "fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
           "traffic_bytes_per_launch": fetch + write, "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
           "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
           "note": "fabric-side L2 counters (Infinity-Cache hits included); FETCH_SIZE doubled per the gfx950 correction"},
          open(out, "w"), indent=1)
print(open(out).read())
