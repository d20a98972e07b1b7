#!/bin/bash
# PMC passes of the n = 4096 and NiCorrectKeyProof kernels at LONG launch shapes (many claims per wavefront): are their cycles per
# instruction at the short shapes a tail artifact?   bash tools/dev/pmc_big_shapes.sh
OUT=gpurun_out/pmc_r03_big
mkdir -p $OUT
for shape in enc4096b1024 ck2048full; do bash profiles/collect_pmc.sh $shape $OUT/$shape > $OUT/collect_$shape.log 2>&1; done
python profiles/aggregate_pmc.py $OUT/enc4096b1024 "k_enc<8, true" --calib profiles/r03_pmc_calibration.json > $OUT/r03_pmc_enc4096_b1024.json
python profiles/aggregate_pmc.py $OUT/ck2048full "k_ck_check<2" --calib profiles/r03_pmc_calibration.json > $OUT/r03_pmc_ck2048_b65536.json
python - <<'PY'
import json
for f in ("gpurun_out/pmc_r03_big/r03_pmc_enc4096_b1024.json", "gpurun_out/pmc_r03_big/r03_pmc_ck2048_b65536.json"):
    for k, r in json.load(open(f)).items():
        if "_derived" in r:
            d = r["_derived"]; print(f.split("/")[-1], {kk.split(" ")[0]: round(d[kk], 3) for kk in ("simd_cycles_per_valu_instr", "valu_active_fraction_of_wave_cycles", "effective_clock_ghz", "hbm_bytes_per_modexp") if kk in d})
PY
