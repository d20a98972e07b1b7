"""kernel time of the two ladders at 4096-bit moduli (one shared modulus), 2048-bit exponents: shared exponent (sliding windows, k_modexp<4,true>)
against per-item exponents (fixed windows, k_modexp<4,false>): ZKP_HIP_LIB=... python tools/dev/perf_ladders.py"""
import json, os, sys
import torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import helpers as H
zkp = H.zkp
ctx = zkp.Context(0); dev = torch.device("cuda", 0); ctx.set_geometry(36)
def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)
B = int(os.environ.get("B", 65536))
for mb, G in ((4096, 4), (2048, 2)):
    w = mb // 32
    N = rnd((1, w), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
    base = rnd((B, w), 3); base[:, -1] &= 0x3FFFFFFF
    out = torch.zeros_like(base)
    for shared in (True, False):
        e = rnd((1 if shared else B, 64), 4); e[:, -1] |= -2**31
        for rep in range(2):
            ctx.timing_reset(True)
            ctx.modexp(mb, 2048, B, base, e, 0 if shared else 64, N, 0, out); ctx.synchronize()
            kms, launches, me = ctx.timing_get(); ctx.timing_reset(False)
        print(json.dumps({"lib": os.path.basename(os.environ.get("ZKP_HIP_LIB", "default")), "mod_bits": mb, "shared_exp": shared, "B": B, "kernel_ms": round(kms, 2)}), flush=True)
