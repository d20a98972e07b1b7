#!/bin/bash
# everything the round's evidence is made of, on one box: bash tools/dev/run_final_evidence.sh [tag]
TAG=${1:-final}
ROUND=${ROUND:-r06}
R=$PWD
# (the whole output with every duration: profiles/${ROUND}/pytest_gpu_durations.txt is what tests/test_gpu_suite_budget.py reads)
timeout 2400 python -m pytest tests -m gpu -q -x --durations=0 --durations-min=1.0 > gpurun_out/${ROUND}_pytest_gpu_durations_$TAG.txt 2>&1
tail -3 gpurun_out/${ROUND}_pytest_gpu_durations_$TAG.txt
python bench.py > gpurun_out/bench_${ROUND}_$TAG.json 2> gpurun_out/bench_${ROUND}_$TAG.err; echo bench rc=$?
(cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${ROUND}_$TAG -- python $R/bench.py --cpu-sample 0 --no-pcie-leg --no-host-api-leg --no-capi-multi-leg --other-reps 1 > $R/gpurun_out/prof_${ROUND}_$TAG.log 2>&1; echo rocprof rc=$?)
cp "$(ls -S $(find gpurun_out/prof_${ROUND}_$TAG -name "*kernel_stats.csv") | head -1)" gpurun_out/${ROUND}_kernel_stats_bench_$TAG.csv      # (the bench process: the largest table)
# the per-dispatch durations of the headline kernel (the stats file averages verify, prove and warm-up launches of one kernel name together)
python - <<PY
import csv, glob, json
rows = []
for f in glob.glob("gpurun_out/prof_${ROUND}_$TAG/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_enc_basen<2>" in r.get("Kernel_Name", ""):
            rows.append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
json.dump({"kernel": "k_enc_basen<2>", "launch_ms_in_dispatch_order": rows,
           "note": "rocprofv3 --kernel-trace of the default bench run: the ~1.2 s launches are the timed verify steps (roofline.kernel_ms_per_launch also holds the ~2 ms of k_expected and k_basen_finish beside them), the ~1.6 s ones the merged c1 + c2 launch of a prove step, the short ones warm-ups / small legs"},
          open("gpurun_out/${ROUND}_kernel_trace_k_enc_basen2_launches_$TAG.json", "w"), indent=1)
PY
find gpurun_out/prof_${ROUND}_$TAG -name "*kernel_trace.csv" -delete
# (SKIP_PMC=1: the counter passes are left out — when the kernels they measure have not changed since the last collection)
if [ -z "$SKIP_PMC" ]; then
  ROUND=$ROUND bash profiles/collect_all.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
  tail -6 gpurun_out/pmc_$TAG.log
fi
