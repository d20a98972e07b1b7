#!/bin/bash
# instruction-cache and wait counters of the dominant Paillier kernel for several library builds on ONE box:
#   bash tools/dev/pmc_icache_ab.sh <outdir> dirA dirB ...      (directories holding libzkp_hip.so + libzkp_hip_lat.so)
R=$PWD
OUT=$1; shift
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$R/$OUT/avail.txt" 2>&1
for d in "$@"; do
  tag=$(basename $d)
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    ZKP_HIP_LIB=$R/$d/libzkp_hip.so ZKP_HIP_LAT_LIB=$R/$d/libzkp_hip_lat.so timeout 600 rocprofv3 --pmc $set --output-format csv -d "$R/$OUT/${tag}_pass$i" -- python $R/bench.py --pmc-shape enc2048 > "$R/$OUT/${tag}_pass$i.log" 2>&1
    echo "$tag pass $i: rc=$?"
  done
done
cd $R
python - "$OUT" <<'P'
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*_pass*"))):
    if not os.path.isdir(d): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            agg[row["Kernel_Name"][:40]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in agg.items():
        if "k_enc" in k:
            print(os.path.basename(d), k, dict(v))
P
