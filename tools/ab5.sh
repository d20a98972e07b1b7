#!/bin/bash
# A/B of whole libraries on ONE box at the headline shape: bash tools/ab5.sh dirA dirB ...   (directories holding libzkp_hip.so + libzkp_hip_lat.so;
# three interleaved repetitions; B = 4096, verify and prove legs, no CPU work)
for rep in 1 2 3; do for d in "$@"; do
  ZKP_HIP_LIB=$PWD/$d/libzkp_hip.so ZKP_HIP_LAT_LIB=$PWD/$d/libzkp_hip_lat.so python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-pcie-leg --no-other-configs --no-host-api-leg --no-capi-multi-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline())
print('$d rep$rep', 'ok' if d['verdicts_ok'] else 'VERDICTS WRONG', 'verify %.1f prove %.1f frac %.4f clock %.3f GHz %.0f W kernel_ms %.1f' % (d['value'], d['prove']['value'], d['roofline']['frac'], d['roofline']['clock']['mean_ghz'], d['roofline']['clock']['mean_power_w'], d['roofline']['kernel_ms_per_launch']))"
done; done
