#!/bin/bash
# tools/pipe_ab.sh: struct-batch pipeline variants through the drop-in API (tools/compat_latency.c), one process per setting, on one box
mkdir -p gpurun_out/r04
run() { echo "== $*"; env "$@" tools/compat_latency_bin | grep batch; }
(
run MOSFHET_HIP_PIPE_PIECE=256
run MOSFHET_HIP_PIPE_PIECE=1024
run MOSFHET_HIP_PIPE_PIECE=512
run MOSFHET_HIP_PIPE_PIECE=128
run MOSFHET_HIP_PIPE_PIECE=256 MOSFHET_HIP_MARSHAL_THREADS=7
run MOSFHET_HIP_PIPE_PIECE=1024 MOSFHET_HIP_MARSHAL_THREADS=7
run MOSFHET_HIP_PIPE_PIECE=256
run MOSFHET_HIP_PIPE_PIECE=1024
) > gpurun_out/r04/pipe_ab2.txt 2>&1
cat gpurun_out/r04/pipe_ab2.txt
