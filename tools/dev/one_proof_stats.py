"""one RangeProofNi (BASELINE configs[0]) proved and verified N times on host buffers: per-call milliseconds, min / median / max of each leg.
python tools/dev/one_proof_stats.py [calls] [proofs]     (ZKP_HIP_LAT_LIB selects a build of the latency engine for A/B runs; ZKP_FUSE_HASH=0: the transcript hash of a verify in a launch of its own)"""
import importlib, json, os, statistics, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers as H
zkp = H.zkp
synth = importlib.import_module("zk-paillier_amd.synth")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
pbh = pb.to(None); wth = wt.to(None); v = np.zeros(B, np.uint8)
for _ in range(3):
    ctx.range_ni_prove(pbh.struct(), wth.struct(), None, None, None, device=False); ctx.range_ni_verify(pbh.struct(), v, device=False)
pv, vv = [], []
for _ in range(calls):
    t0 = time.perf_counter(); ctx.range_ni_prove(pbh.struct(), wth.struct(), None, None, None, device=False)
    t1 = time.perf_counter(); ctx.range_ni_verify(pbh.struct(), v, device=False)
    t2 = time.perf_counter()
    pv.append(1e3 * (t1 - t0)); vv.append(1e3 * (t2 - t1))
st = lambda x: {"min": round(min(x), 3), "median": round(statistics.median(x), 3), "max": round(max(x), 3), "max_over_min": round(max(x) / min(x), 3)}
print(json.dumps({"proofs": B, "calls": calls, "lat_lib": os.environ.get("ZKP_HIP_LAT_LIB", "in-tree"), "fuse_hash_env": os.environ.get("ZKP_FUSE_HASH", "unset"), "fused": ctx.last_fused_hash(), "accepted": bool(v.all()), "geometry": ctx.last_geometry(),
                  "prove_ms": st(pv), "verify_ms": st(vv), "verify_all": [round(x, 2) for x in vv]}))
