#!/usr/bin/env python3
"""Small-batch latency of RangeProofNi prove / verify (host buffers in and out, what one caller of the crate sees), by hand on a
GPU box: python tests/perf_gpu_latency.py [n_bits].  ZKP_HIP_LIB selects the build; the checksum lets two builds be compared."""
import hashlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H
import importlib
zkp = H.zkp
synth = importlib.import_module("zk-paillier_amd.synth")
n_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = zkp.Context(0)
dev = torch.device("cuda", 0)
lpl = zkp.load().zkp_build_limbs_per_lane()
nkey = synth.BENCH_N if n_bits == 2048 else synth.bench_key_4096()[2]
lat = ctx.latency_limbs_per_lane()
geometries = [lpl, lat, 0] if lat else [lpl]          # pinned to each engine, then the library's own choice
for B in (1, 2, 8, 32, 128, 512):
    pb, wt = synth.synth_range_inputs(nkey, n_bits, B, seed=7, device=dev)
    ctx.set_geometry(lpl)
    ctx.paillier_enc(n_bits, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    wth = wt.to(None)
    for geom in geometries:
        ctx.set_geometry(geom)
        pbh = pb.to(None)
        v = np.zeros(B, np.uint8)
        best_p = best_v = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            ctx.range_ni_prove(pbh.struct(), wth.struct(), None, None, None, device=False)
            t1 = time.perf_counter()
            ctx.range_ni_verify(pbh.struct(), v, device=False)
            t2 = time.perf_counter()
            best_p = min(best_p, t1 - t0); best_v = min(best_v, t2 - t1)
        h = hashlib.sha256()
        for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
            h.update(np.ascontiguousarray(getattr(pbh, f)).tobytes())
        print(json.dumps({"what": "RangeProofNi", "n_bits": n_bits, "B": B, "geometry": geom or "automatic", "ran_on_limbs_per_lane": ctx.last_geometry(),
                          "prove_ms": 1e3 * best_p, "verify_ms": 1e3 * best_v, "accepted": bool((v == 1).all()), "sha": h.hexdigest()[:16]}), flush=True)
ctx.set_geometry(0)


# ---- CompositeDLogProof and NiCorrectKeyProof verification at protocol-sized batches, both engines (device buffers)
def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)


def best_of(fn, reps=3):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


if n_bits == 2048:
    for B in (1, 16, 256, 4096):
        N = rnd((B, 64), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
        g_ = rnd((B, 64), 3); g_[:, -1] &= 0x3FFFFFFF
        ni = rnd((B, 64), 4); ni[:, -1] &= 0x3FFFFFFF
        x = rnd((B, 64), 5); x[:, -1] &= 0x3FFFFFFF
        y = rnd((B, 24), 6); y[:, 17:] = 0
        sg = rnd((B, 11, 64), 7); sg[:, :, -1] &= 0x3FFFFFFF
        v = torch.zeros(B, dtype=torch.uint8, device=dev)
        rec = {"B": B}
        for geom in geometries:
            ctx.set_geometry(geom)
            rec[f"dlog_verify_ms_w{geom}"] = 1e3 * best_of(lambda: ctx.dlog_verify(2048, 768, B, N, g_, ni, x, y, v))
            rec[f"dlog_ran_on_w{geom}"] = ctx.last_geometry()
            rec[f"correct_key_verify_ms_w{geom}"] = 1e3 * best_of(lambda: ctx.correct_key_ni_verify(2048, B, N, sg, b"KZen", v))
            rec[f"ck_ran_on_w{geom}"] = ctx.last_geometry()
        ctx.set_geometry(0)
        print(json.dumps(rec), flush=True)
