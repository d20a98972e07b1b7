#!/bin/bash
# PMC passes of ONE dominant-kernel shape (run on the GPU box from the repo root):
#     bash profiles/collect_pmc.sh <shape> <outdir>        shape = enc2048 | enc2048full | enc2048keys | enc4096 | ck2048 | tabread  (bench.py --pmc-shape)
# tabread = the calibration launches (a known number of bytes read, then written, in the table access pattern): aggregate_pmc.py --calibrate
# Counter sets are collected in separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides
# the kernel trace are combined with --pmc).  Aggregate with profiles/aggregate_pmc.py.
set -u
R=$PWD
SHAPE=${1:-enc2048}
OUT=${2:-gpurun_out/pmc_$SHAPE}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --pmc-shape $SHAPE"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d "$R/$OUT/pass$i" -- $CMD > "$R/$OUT/pass$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done
grep -h pmc_shape "$R/$OUT"/pass1.log | tail -1 > "$R/$OUT/shape.json"
