#!/bin/bash
# tools/two_procs_probe.sh: two processes bootstrapping lvl2 batches on ONE GPU at the same time, with and without the per-XCD rendezvous (what does the bounded wait cost
# when the teams of a launch are NOT all resident?)
for P in 0 32; do
  echo "== MOSFHET_HIP_PACE=$P, two processes at once"
  MOSFHET_HIP_PACE=$P python tools/gpu_perf.py 4096 lvl2 > /tmp/p1.txt 2>&1 &
  A=$!
  MOSFHET_HIP_PACE=$P python tools/gpu_perf.py 4096 lvl2 > /tmp/p2.txt 2>&1 &
  B=$!
  wait $A; wait $B
  tail -1 /tmp/p1.txt; tail -1 /tmp/p2.txt
done
echo "== one process alone"
python tools/gpu_perf.py 4096 lvl2 2>&1 | tail -1
