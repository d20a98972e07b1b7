#!/bin/bash
# kernel trace (with timestamps: the two-stream overlap of a one-proof verify) and one PMC pass of five one-proof prove + verify calls
set -u
R=$PWD
OUT=${1:-gpurun_out/one_proof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace" -o t -- python "$R/tools/dev/lat1.py" > "$R/$OUT/trace.log" 2>&1; echo "trace rc=$?"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$R/$OUT/pmc" -o p -- python "$R/tools/dev/lat1.py" > "$R/$OUT/pmc.log" 2>&1; echo "pmc rc=$?"
find "$R/$OUT" -name "*.csv" | head
