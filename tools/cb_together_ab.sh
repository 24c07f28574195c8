#!/bin/bash
# tools/cb_together_ab.sh: circuit_bootstrap_3 at small batches with one packing switch per level (MOSFHET_HIP_CB_TOGETHER=0) or one for all levels (1), same box
for B in 1 16 128 256 384 1024; do
  for T in 0 1; do
    echo "== B=$B MOSFHET_HIP_CB_TOGETHER=$T"
    MOSFHET_HIP_CB_TOGETHER=$T python tools/gpu_perf_cb.py $B 2>&1 | grep "circuit_bootstrap_3"
  done
done
