"""Times one Paillier Enc launch of `count` items on the latency engine's base-n kernel k_enc_basen<8> (8 Enc per wavefront; 8192 items = one wavefront
per SIMD = 32 proofs' worth), results unchecked.  python tools/dev/basen8_time.py [counts ...]"""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zkp = importlib.import_module("zk-paillier_amd")
synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0)
dev = torch.device("cuda", 0)
rec = {"lat_lib": os.environ.get("ZKP_HIP_LAT_LIB", "in-tree")}
ctx.set_geometry(9); ctx.set_enc_form("basen"); ctx.set_r2l(0)
for count in [int(v) for v in sys.argv[1:]] or [6144, 8192, 16384]:
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, count, seed=7, device=dev)
    best = 1e9
    for rep in range(5):
        ctx.synchronize(); t0 = time.perf_counter()
        ctx.paillier_enc(2048, count, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    lanes, ok = ctx.diag_basen_last()
    rec[f"count_{count}_ms"] = round(1e3 * best, 3)
    rec["ran"] = f"w{ctx.last_geometry()} lanes={lanes} ok={ok}"
print(json.dumps(rec), flush=True)
