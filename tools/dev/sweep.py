import os, sys, time, importlib, json
import numpy as np, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import helpers as H
zkp = H.zkp
synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)
def best_of(fn, reps=3):
    fn(); ctx.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
for B in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    N = rnd((B, 64), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
    sg = rnd((B, 11, 64), 7); sg[:, :, -1] &= 0x3FFFFFFF
    g_ = rnd((B, 64), 3); g_[:, -1] &= 0x3FFFFFFF
    ni = rnd((B, 64), 4); ni[:, -1] &= 0x3FFFFFFF
    x = rnd((B, 64), 5); x[:, -1] &= 0x3FFFFFFF
    y = rnd((B, 24), 6); y[:, 17:] = 0
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"B": B}
    for geom in (36, 9):
        ctx.set_geometry(geom)
        rec[f"ck_w{geom}"] = round(best_of(lambda: ctx.correct_key_ni_verify(2048, B, N, sg, b"KZen", v)), 2)
        rec[f"dlog_w{geom}"] = round(best_of(lambda: ctx.dlog_verify(2048, 768, B, N, g_, ni, x, y, v)), 2)
    print(json.dumps(rec), flush=True)
for B in (16, 32, 48, 64, 96, 128):
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
    ctx.set_geometry(36)
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"range_B": B}
    for geom in (36, 9):
        ctx.set_geometry(geom)
        rec[f"prove_w{geom}"] = round(best_of(lambda: ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True)), 2)
        rec[f"verify_w{geom}"] = round(best_of(lambda: ctx.range_ni_verify(pb.struct(), v, device=True)), 2)
    print(json.dumps(rec), flush=True)
