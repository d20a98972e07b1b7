"""RangeProofNi prove / verify (device-resident, n = 2048, one key) from 1 to 1024 proofs under every kernel family the library has — the
throughput engine (36 limbs per lane) and the latency engine (9), each with the n^2-sized kernels and in base-n form — and under the
library's own choice.  One JSON line per size: the table behind the routing rules of csrc/zkp_api.hip (route_latency, launch_basen).
python tools/dev/size_sweep.py [sizes ...]"""
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
sizes = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 512, 768, 1024]
LAT = ctx.latency_limbs_per_lane()          # 9 as shipped; $ZKP_HIP_LAT_LIB may name another build of the secondary engine (A/B runs)
MID = ctx.mid_limbs_per_lane()
FAMILIES = (("w36_n2", 36, "n2"), ("w36_basen", 36, "basen")) + ((("w18_basen", 18, "basen"),) if MID == 18 else ()) + ((f"w{LAT}_n2", LAT, "n2"), (f"w{LAT}_basen", LAT, "basen"), ("auto", 0, "auto"))
if os.environ.get("ZKP_SWEEP_AUTO_ONLY") == "1":      # the library's own choice only (A/B of whole engine sets: $ZKP_HIP_MID_LIB=/nonexistent removes the mid engine)
    FAMILIES = (("auto", 0, "auto"),)


def best_of(fn, reps=3):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(1e3 * best, 2)


for B in sizes:
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
    ctx.set_geometry(0); ctx.set_enc_form("auto")
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"B": B, "mid_engine": MID}
    for name, geom, form in FAMILIES:
        if geom == LAT and B > 256 or (name == "w36_basen" and B < 8):
            continue
        ctx.set_geometry(geom); ctx.set_enc_form(form)
        rec[name] = [best_of(lambda: ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True)),
                     best_of(lambda: ctx.range_ni_verify(pb.struct(), v, device=True))]
        if name == "auto":
            lanes, ok = ctx.diag_basen_last()
            rec["auto_ran"] = f"w{ctx.last_geometry()}_" + ("r2l" if ctx.r2l_last() else "basen" if lanes and ok else "n2")
        assert bool(v.all()), (B, name)
    print(json.dumps(rec), flush=True)
