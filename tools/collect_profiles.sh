#!/bin/bash
# tools/collect_profiles.sh [tag]: copy what tools/profile_r04.sh left under gpurun_out/ into profiles/ (summaries, the kernel-stats table each summary names,
# the three measured-traffic files bench.py reads).  Run here after the gpurun call that profiled THIS build (the traffic files carry the library's source hash).
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for pair in pbs:pbs ep:ep ep2:ep_lvl2 lvl2:lvl2 ks:ks cb:cb unf:unf split:split; do
  src=${pair%%:*}; dst=${pair##*:}
  d=gpurun_out/prof_${TAG}_$src
  [ -f $d/summary.txt ] || { echo "no $d/summary.txt"; continue; }
  cp $d/summary.txt profiles/${TAG}_${dst}_summary.txt
  ks=$(grep -o "trace/[^ )]*kernel_stats.csv" $d/summary.txt | head -1)
  [ -n "$ks" ] && cp $d/$ks profiles/${TAG}_${dst}_kernel_stats.csv
done
python tools/make_traffic_json.py profiles/${TAG}_pbs_summary.txt profiles/latest_traffic.json "pbs_kernel<mosfhet::Fft1024, 2, 8>" 4096 > /dev/null
python tools/make_traffic_json.py profiles/${TAG}_ep_summary.txt profiles/latest_traffic_ep.json "external_product_ldskey_kernel<2, 8" 65536 > /dev/null
python tools/make_traffic_json.py profiles/${TAG}_ep_lvl2_summary.txt profiles/latest_traffic_ep_lvl2.json "external_product_kernel<mosfhet::Fft" 16384 > /dev/null
grep -h "srchash\|traffic_bytes" profiles/latest_traffic*.json
