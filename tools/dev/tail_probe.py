#!/usr/bin/env python3
"""Is a modexp-class launch quantised in rounds of (resident wavefronts x items per wavefront)?  Times zkp_paillier_enc_batch (shared key,
n = 2048: k_enc<4, true>, 16 items per wavefront, 2048 resident wavefronts = 32768 items per round) at counts just below / at / just
above a whole number of rounds.  If +16 items cost a whole round (~77 ms), the last partial round of a launch is worth scheduling
differently (the headline verify launch is 24.002 rounds).   python tools/dev/tail_probe.py > gpurun_out/r04/tail_probe.jsonl"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
zkp = H.zkp
ctx = zkp.Context(0)
ctx.set_geometry(zkp.load().zkp_build_limbs_per_lane())
dev = torch.device("cuda", 0)
kw = 64
n = torch.from_numpy(H.L.int_to_limbs(H.fixture_key()[2], kw).view(np.int32)).to(dev).reshape(1, kw)
ROUND = 32768
big = 8 * ROUND + 8192
g = torch.Generator(device=dev); g.manual_seed(5)
m = torch.randint(0, 2**31 - 1, (big, kw), dtype=torch.int32, device=dev, generator=g); m[:, 8:] = 0
r = torch.randint(0, 2**31 - 1, (big, kw), dtype=torch.int32, device=dev, generator=g); r[:, -1] &= 0x3FFFFFFF
out = torch.zeros((big, 2 * kw), dtype=torch.int32, device=dev)
for count in (4 * ROUND, 4 * ROUND, 4 * ROUND - 16, 4 * ROUND + 16, 4 * ROUND + 58, 4 * ROUND + 2048, 4 * ROUND + 8192, 4 * ROUND + 16384, 5 * ROUND, 8 * ROUND, 8 * ROUND + 16):
    ctx.timing_reset(True)
    ctx.paillier_enc(2048, count, n, 0, m[:count], r[:count], out[:count]); ctx.synchronize()
    kms, launches, me = ctx.timing_get(); ctx.timing_reset(False)
    print(json.dumps({"items": count, "rounds": count / ROUND, "kernel_ms": round(kms, 2), "ms_per_whole_round": round(kms / max(1, count // ROUND), 2)}), flush=True)
