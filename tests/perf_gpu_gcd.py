#!/usr/bin/env python3
"""Timing of the latency-class kernels with device-resident buffers (by hand on a GPU box: python tests/perf_gpu_gcd.py):
zkp_modinv_batch at 4096 bits and CompositeDLogProof verify at N = 2048 for B = 4096 and 65536.  Prints one JSON line each."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from helpers import pm, L

zkp = H.zkp
ctx = zkp.Context(0)
dev = torch.device("cuda", 0)


def rnd(shape, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)


def timeit(fn, reps=3):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


# ---- mod_inv, 4096-bit shared modulus n^2 (fixture key)
p, q, n = H.fixture_key()
nn = n * n
kw = 128
m = torch.from_numpy(L.int_to_limbs(nn, kw).view(np.int32)).to(dev).view(1, kw)
for B in (4096, 16384):
    a = rnd((B, kw), 1); a[:, -1] &= 0x0FFFFFFF          # a < 2^4092 < n^2
    out = torch.zeros_like(a); st = torch.zeros(B, dtype=torch.uint8, device=dev)
    dt = timeit(lambda: ctx.modinv(4096, B, a, m, 0, out, st))
    print(json.dumps({"what": "zkp_modinv_batch 4096-bit", "batch": B, "ms": 1e3 * dt, "inverses_per_s": B / dt, "ok": int((st == 0).sum().item())}), flush=True)

# ---- CompositeDLogProof verify, N = 2048 (random odd pseudo-moduli: the two gcds, two modexps, hash and compare all run; verdicts are rejects)
for B in (4096, 65536):
    N = rnd((B, 64), 2); N[:, 0] |= 1; N[:, -1] |= -2**31
    g_ = rnd((B, 64), 3); g_[:, -1] &= 0x3FFFFFFF
    ni = rnd((B, 64), 4); ni[:, -1] &= 0x3FFFFFFF
    x = rnd((B, 64), 5); x[:, -1] &= 0x3FFFFFFF
    y = rnd((B, 24), 6); y[:, 17:] = 0
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    dt = timeit(lambda: ctx.dlog_verify(2048, 768, B, N, g_, ni, x, y, v))
    print(json.dumps({"what": "zkp_dlog_verify_batch N=2048", "batch": B, "ms": 1e3 * dt, "verifies_per_s": B / dt}), flush=True)
