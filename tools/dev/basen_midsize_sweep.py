"""RangeProofNi prove / verify at mid-size batches on the throughput engine: run twice, with ZKP_BASEN=0 and without, to see where the
base-n kernels (32 Enc per wavefront: half the wavefronts of the n^2-sized launch) stop paying.  python tools/dev/basen_midsize_sweep.py"""
import os, sys, time, importlib, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers as H
zkp = H.zkp
synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
ctx.set_geometry(36)
def best_of(fn, reps=3):
    fn(); ctx.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
for B in (48, 64, 96, 128, 160, 192, 256, 320, 384, 512, 768, 1024):
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"basen": os.environ.get("ZKP_BASEN", "default"), "B": B,
           "prove_ms": round(best_of(lambda: ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True)), 2),
           "verify_ms": round(best_of(lambda: ctx.range_ni_verify(pb.struct(), v, device=True)), 2)}
    print(json.dumps(rec), flush=True)
