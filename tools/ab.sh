#!/bin/bash
# A/B of kernel variants on ONE box: bash tools/ab.sh lib1.so lib2.so ...   (two interleaved repetitions)
# ZKP_AB_BIG=128 adds the n = 4096 leg (k_enc<8, true>) with that many proofs
BIG=${ZKP_AB_BIG:-0}
for rep in 1 2; do for lib in "$@"; do
  ZKP_HIP_LIB=$PWD/$lib python bench.py --batch 2048 --steps 2 --warmup 1 --no-prove-leg --cpu-sample 0 --no-pcie-leg --big-batch $BIG --distinct-batch 1024 --interactive-batch 0 --other-reps 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); o=d['other_configs']
ck=[v for k,v in o.items() if 'configs[3]' in k][0]; di=[v for k,v in o.items() if 'DISTINCT' in k][0]; big=[v for k,v in o.items() if 'configs[4]' in k]
print('$lib rep$rep', 'verify %.1f frac %.4f' % (d['value'], d['roofline']['frac']), 'ck frac %.4f' % ck['frac'], 'distinct verify frac %.4f prove frac %.4f' % (di['roofline']['frac'], di['prove_frac']),
      ('n4096 verify frac %.4f prove frac %.4f' % (big[0]['roofline']['frac'], big[0]['prove_frac'])) if big else '')"
done; done
