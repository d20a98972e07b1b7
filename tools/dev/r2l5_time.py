"""Times ONE Paillier Enc launch of `count` items (default 256: one proof's prove launch) on the latency engine's right-to-left ladder, five
wavefronts per Enc and one, results unchecked (the probe variants of tools/dev/r2l5_variants.sh compute garbage).  python tools/dev/r2l5_time.py [count]"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zkp = importlib.import_module("zk-paillier_amd")
synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0)
dev = torch.device("cuda", 0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, count, seed=7, device=dev)
rec = {"lat_lib": os.environ.get("ZKP_HIP_LAT_LIB", "in-tree"), "count": count}
ctx.set_geometry(9); ctx.set_r2l(2)
for lanes in (36, 12):
    ctx.set_r2l_lanes(lanes)
    best = 1e9
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.paillier_enc(2048, count, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    rec[f"lanes_{lanes}_ms"] = round(1e3 * best, 3)
    rec[f"lanes_{lanes}_ran"] = ctx.r2l_lanes_last()
print(json.dumps(rec), flush=True)
