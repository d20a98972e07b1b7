#!/bin/bash
# A/B of kernel variants on ONE box: bash scratch/ab.sh lib1.so lib2.so ...   (two interleaved repetitions)
for rep in 1 2; do for lib in "$@"; do
  ZKP_HIP_LIB=$PWD/$lib python bench.py --batch 2048 --steps 2 --warmup 1 --no-prove-leg --cpu-sample 0 --no-pcie-leg --big-batch 0 --distinct-batch 1024 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); o=d['other_configs']; ks=list(o); print('$lib rep$rep', 'verify %.1f frac %.4f' % (d['value'], d['roofline']['frac']), 'ck frac %.4f' % o[ks[1]]['frac'], 'distinct verify frac %.4f prove frac %.4f' % (o[ks[2]]['roofline']['frac'], o[ks[2]]['prove_frac']))"
done; done
