#!/bin/bash
# FUNCTIONAL check of `bench.py --gpus 2` (both scalings, every sharded leg, unequal blocks) on a box with ONE GPU:
# both ranks use cuda:0, the collectives run over gloo.  Timings are meaningless; rc and verdicts_ok are the result.
export ZKP_BENCH_SHARED_GPU=1
for scaling in weak strong; do
  python bench.py --gpus 2 --scaling $scaling --batch 130 --steps 1 --warmup 0 --cpu-sample 0 --no-pcie-leg --big-batch 33 --distinct-batch 66 --ck-batch 2049 --interactive-batch 65 --other-reps 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline())
print('$scaling', 'n_gpus', d['n_gpus'], 'ok', d['verdicts_ok'], 'value %.1f' % d['value'], d['config']['proofs_per_rank'], d['config']['proofs_total'])
print('    rccl', d['rccl']['backend'], 'ranks_seen', d['rccl']['ranks_seen'], 'boards', d['rccl']['distinct_boards_seen'])
for m, v in d['scaling_values'].items(): print('    scaling_values', m, v['proofs_total'], v['proofs_per_rank'], 'verify %.1f/s prove %.1f/s' % (v['verifies_per_s'], v['proofs_per_s']), v['verify_phases_rank0'])
for k,v in d['other_configs'].items(): print('   ', k[:50], {kk: v[kk] for kk in ('n_gpus','batch_per_rank','keys_per_rank','verdicts_ok','all_rejected_as_expected','all_accepted') if kk in v})"
  echo "rc=$?"
done
