#!/bin/bash
# latency engine under another machine-scheduler strategy: ZKP_HIP_LAT_LIB selects the build
for l in "" ablibs/lat_maxilp.so ablibs/lat_default.so "" ablibs/lat_maxilp.so ablibs/lat_default.so; do
  if [ -n "$l" ]; then export ZKP_HIP_LAT_LIB=$PWD/$l; else unset ZKP_HIP_LAT_LIB; fi
  python tests/perf_gpu_latency.py 2048 2>/dev/null | grep RangeProofNi | python -c "
import sys,json
out=[]
for line in sys.stdin:
    d=json.loads(line)
    if d['geometry']==9 and d['B'] in (1,8,32): out.append('B%d %.1f/%.1f' % (d['B'], d['prove_ms'], d['verify_ms']))
print('${l:-head}', ' '.join(out))"
  python tools/dev/lat_sqr_ab.py 2>/dev/null | python -c "
import sys,json
print('   ', ' | '.join('B%d dlog %.2f ck %.2f' % (d['B'], d['dlog_verify_ms'], d['correct_key_verify_ms']) for d in map(json.loads, sys.stdin)))"
done
