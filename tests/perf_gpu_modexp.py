#!/usr/bin/env python3
"""Per-product time of k_modexp<2, false> (2048-bit moduli, per-item exponents) against exponent length and batch size, by hand on a GPU
box: python tests/perf_gpu_modexp.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
zkp = H.zkp
ctx = zkp.Context(0)
dev = torch.device("cuda", 0)
def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)
for B in (65536, 131072, 32768):
    N = rnd((B, 64), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
    base = rnd((B, 64), 3); base[:, -1] &= 0x3FFFFFFF
    out = torch.zeros_like(base)
    for eb in (256, 512, 768, 1024, 2048):
        e = rnd((B, eb // 32), 4)
        ctx.modexp(2048, eb, B, base, e, eb // 32, N, 64, out); ctx.synchronize()
        ctx.timing_reset(True)
        ctx.modexp(2048, eb, B, base, e, eb // 32, N, 64, out); ctx.synchronize()
        kms, launches, me = ctx.timing_get(); ctx.timing_reset(False)
        products = eb * 1.2 + 30
        print(json.dumps({"B": B, "exp_bits": eb, "kernel_ms": kms, "us_per_product": 1e3 * kms / products, "modexp_per_s": B / (kms * 1e-3)}), flush=True)
