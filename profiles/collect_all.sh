#!/bin/bash
# PMC evidence of a round (ROUND=r04 by default) in one go (run on the GPU box from the repo root): ROUND=r04 bash profiles/collect_all.sh [libtag]
#   1. tabread: calibration of FETCH_SIZE / WRITE_SIZE on a known number of bytes in the table access pattern
#   2. the dominant kernels at LONG launch shapes (enc2048full = B 4096, the full headline shape; 1024 proofs at n = 4096; 65 536 keys):
#      short launches count the idle tail of the grid as cycles
#      (Paillier Enc runs in base-n form since the middle of round 4: k_enc_basen<2> / <4>, k_enc_basen_keys<2> under per-proof keys;
#      ZKP_BASEN=0 in the environment brings the n^2-sized k_enc<4, true> / k_enc<8, true> / k_enc<4, false> back for a comparison pass)
#   3. aggregation with the calibrated factors and the effective clock -> gpurun_out/pmc_${ROUND}_<tag>/*.json (copy into profiles/)
TAG=${1:-final}
ROUND=${ROUND:-r06}
OUT=gpurun_out/pmc_${ROUND}_$TAG
mkdir -p $OUT
for shape in tabread enc2048full enc2048keys enc4096b1024 ck2048full; do
  bash profiles/collect_pmc.sh $shape $OUT/$shape > $OUT/collect_$shape.log 2>&1
done
python profiles/aggregate_pmc.py --calibrate $OUT/tabread > $OUT/${ROUND}_pmc_calibration.json
python profiles/aggregate_pmc.py $OUT/enc2048full "k_enc_basen<2>" --calib $OUT/${ROUND}_pmc_calibration.json > $OUT/${ROUND}_pmc_${TAG}_enc2048_shared_b4096.json
python profiles/aggregate_pmc.py $OUT/enc2048keys "k_enc_basen_keys<2>" --calib $OUT/${ROUND}_pmc_calibration.json > $OUT/${ROUND}_pmc_${TAG}_enc2048_keys.json
python profiles/aggregate_pmc.py $OUT/enc4096b1024 "k_enc_basen<4>" --calib $OUT/${ROUND}_pmc_calibration.json > $OUT/${ROUND}_pmc_${TAG}_enc4096_b1024.json
python profiles/aggregate_pmc.py $OUT/ck2048full "k_ck_check<2" --calib $OUT/${ROUND}_pmc_calibration.json > $OUT/${ROUND}_pmc_${TAG}_ck2048_b65536.json
ls -la $OUT/*.json
