#!/bin/bash
# A/B of an ENVIRONMENT switch of the library on ONE box at the headline shape: bash tools/dev/ab_env.sh "VAR=a" "VAR=b" ...   (three interleaved repetitions)
for rep in 1 2 3; do for e in "$@"; do
  env $e python bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-pcie-leg --no-other-configs --no-host-api-leg --no-capi-multi-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline())
print('$e rep$rep', 'ok' if d['verdicts_ok'] else 'VERDICTS WRONG', 'verify %.1f (%.1f ms/step) prove %.1f frac %.4f clock %.3f GHz kernel_ms %.1f' % (d['value'], d['ms_per_step'], d['prove']['value'], d['roofline']['frac'], d['roofline']['clock']['mean_ghz'], d['roofline']['kernel_ms_per_launch']))"
done; done
