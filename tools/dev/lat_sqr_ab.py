"""latency engine with / without montsqr on its window ladders (2048-bit moduli: NiCorrectKeyProof, CompositeDLogProof), engine pinned to W = 9:
ZKP_HIP_LAT_LIB=... python tools/dev/lat_sqr_ab.py"""
import json, os, sys, time
import torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import helpers as H
zkp = H.zkp
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)
def best_of(fn, reps=3):
    fn(); ctx.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    return round(1e3 * best, 2)
ctx.set_geometry(9)
for B in (256, 1024, 4096, 16384):
    N = rnd((B, 64), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
    g_ = rnd((B, 64), 3); g_[:, -1] &= 0x3FFFFFFF
    ni = rnd((B, 64), 4); ni[:, -1] &= 0x3FFFFFFF
    x = rnd((B, 64), 5); x[:, -1] &= 0x3FFFFFFF
    y = rnd((B, 24), 6); y[:, 17:] = 0
    sg = rnd((B, 11, 64), 7); sg[:, :, -1] &= 0x3FFFFFFF
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    print(json.dumps({"lat_lib": os.path.basename(os.environ.get("ZKP_HIP_LAT_LIB", "default")), "B": B,
                      "dlog_verify_ms": best_of(lambda: ctx.dlog_verify(2048, 768, B, N, g_, ni, x, y, v)),
                      "correct_key_verify_ms": best_of(lambda: ctx.correct_key_ni_verify(2048, B, N, sg, b"KZen", v))}), flush=True)
