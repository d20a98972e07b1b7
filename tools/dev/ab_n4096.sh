#!/bin/bash
# A/B of the n = 4096 leg on ONE box: bash tools/dev/ab_n4096.sh lib1.so lib2.so   (two interleaved repetitions, batch 512)
for rep in 1 2; do for lib in "$@"; do
  ZKP_HIP_LIB=$PWD/$lib python bench.py --batch 512 --steps 1 --warmup 1 --no-prove-leg --cpu-sample 0 --no-pcie-leg --big-batch 512 --distinct-batch 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); o=d['other_configs']; k=[x for x in o if 'n=4096' in x][0]; print('$lib rep$rep', 'n4096 verify %.2f prove %.2f frac %.4f prove_frac %.4f' % (o[k]['verifies_per_s'], o[k]['proofs_per_s'], o[k]['roofline']['frac'], o[k]['prove_frac']))"
done; done
