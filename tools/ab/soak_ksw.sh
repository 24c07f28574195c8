#!/bin/bash
# on the GPU box: the word-lane key-switch kernel repeated on the shapes of the configs and on odd ones, every launch compared on the device (tools/ab/ksw_ab.hip soak mode)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r06b
OUT=$ROOT/gpurun_out/r06b/ksw_soak.txt
: > $OUT
B=$ROOT/tools/ab/_build/ksw_ab
for c in "30000 600 100 586 585 5 2 1" "30000 129 64 130 -1 4 4 0" "20000 300 50 200 199 20 2 0" "20000 513 17 641 640 9 4 0" "4000 4096 1024 586 585 5 2 0" "3000 1024 2048 633 632 8 4 0" \
         "1500 4096 2048 633 632 8 4 0" "3000 128 2048 4096 2048 6 4 0" "2000 512 2048 4096 2048 6 4 1" "1500 1024 2048 4096 2048 6 4 0" "1000 1024 2048 4096 2048 6 4 1"; do
  set -- $c
  timeout -k 10 400 $B soak "$@" >> $OUT 2>&1 || echo "FAILED: $c" >> $OUT
done
grep -c "0 differing words in total, 0 wavefronts" $OUT
grep "soak:\|FAILED\|MISMATCH" $OUT
