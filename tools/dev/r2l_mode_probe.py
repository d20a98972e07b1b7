import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
zkp = importlib.import_module("zk-paillier_amd"); synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
def best(fn, reps=4):
    fn(); ctx.synchronize(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); b = min(b, time.perf_counter() - t0)
    return round(1e3 * b, 2)
for B in (8, 9, 10, 11, 12, 14, 16):
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"B": B}
    for mode in (1, 2):
        ctx.set_geometry(0); ctx.set_enc_form("auto"); ctx.set_r2l(mode)
        rec["r2l_mode_%d" % mode] = [best(lambda: ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True)),
                                     best(lambda: ctx.range_ni_verify(pb.struct(), v, device=True))]
        rec["ran_%d" % mode] = "r2l" if ctx.r2l_last() else "other"
        assert bool(v.all())
    ctx.set_r2l(1)
    print(json.dumps(rec), flush=True)
