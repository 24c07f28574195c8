#!/bin/bash
# builds tools/ab/_build/ksw_ab (same-box A/B of the two table key-switch forms) and prints the new kernels' resource usage
set -e
cd "$(dirname "$0")/_build" 2>/dev/null || { mkdir -p "$(dirname "$0")/_build"; cd "$(dirname "$0")/_build"; }
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -save-temps=obj -I ../../../mosfhet_amd/csrc -o ksw_ab ../ksw_ab.hip 2>&1 | grep -v "^note\|reserved registers\|warnings gen" || true
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|name):" ksw_ab-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - - | grep -i "table_ks" | sed 's/\s\+/ /g'
