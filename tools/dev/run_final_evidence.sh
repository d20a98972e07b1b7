#!/bin/bash
# everything the round's evidence is made of, on one box: bash tools/dev/run_final_evidence.sh [tag]
TAG=${1:-final}
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r03_pytest_gpu_$TAG.txt
tail -3 gpurun_out/r03_pytest_gpu_$TAG.txt
python bench.py > gpurun_out/bench_r03_$TAG.json 2> gpurun_out/bench_r03_$TAG.err; echo bench rc=$?
(cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_$TAG -- python $R/bench.py --cpu-sample 0 --no-pcie-leg --other-reps 1 > $R/gpurun_out/prof_r03_$TAG.log 2>&1; echo rocprof rc=$?)
find gpurun_out/prof_r03_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_kernel_stats_bench_$TAG.csv \;
find gpurun_out/prof_r03_$TAG -name "*kernel_trace.csv" -delete
bash profiles/r03_collect_all.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
tail -6 gpurun_out/pmc_$TAG.log
