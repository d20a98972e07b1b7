#!/bin/bash
# PMC passes of the verify leg (run on the GPU box from the repo root: bash profiles/collect_pmc.sh <outdir>).
# Counter sets are collected in separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides
# the kernel trace are combined with --pmc).  Aggregate with profiles/aggregate_pmc.py.
set -u
R=$PWD
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --batch 512 --steps 1 --warmup 0 --cpu-sample 0 --no-prove-leg --no-other-configs"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d "$R/$OUT/pass$i" -- $CMD > "$R/$OUT/pass$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done
