#!/usr/bin/env python3
"""Randomised soak + timing of zkp_modinv_batch, MulProof and CorrectMessageProof against the oracle (by hand on a GPU box:
python tests/soak_gpu_inv.py [rounds]).  Inverse inputs mix uniform values with structured ones (few bits set, long runs
of zero words at either end, multiples of the prime factors, values next to the modulus)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from helpers import pm, L
import oracle_lib


def structured(rng, d, mod, p, q, count):
    bits = mod.bit_length()
    out = []
    for i in range(count):
        k = i % 8
        if k == 0: v = d.below(mod)
        elif k == 1: v = 1 << int(rng.integers(0, bits - 1))
        elif k == 2: v = (d.below(mod) >> int(rng.integers(1, bits - 2)))
        elif k == 3: v = (d.below(mod) >> int(rng.integers(32, 400))) << int(rng.integers(32, 400))
        elif k == 4: v = (p if rng.random() < 0.5 else q) * d.below(1 << 200)
        elif k == 5: v = mod - 1 - d.below(1 << int(rng.integers(1, 300)))
        elif k == 6: v = sum(1 << int(x) for x in rng.integers(0, bits - 1, size=int(rng.integers(1, 6))))
        else: v = d.below(1 << int(rng.integers(1, bits)))
        out.append(v % mod)
    return out


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    zkp = H.zkp
    ctx = zkp.Context(0)
    oracle = oracle_lib.Oracle()
    oracle.set_threads(min(16, oracle.max_threads()))
    total = 0
    for rd in range(rounds):
        rng = np.random.default_rng(77 + rd)
        for mod_bits, key_bits in ((2048, 1024), (4096, 2048)):
            kw = mod_bits // 32
            p, q, n = H.test_key(key_bits, tag=rd % 2) if key_bits != 2048 else H.fixture_key()
            nn = n * n
            d = pm.Drbg(b"soak-inv-%d-%d" % (rd, mod_bits))
            vals = structured(rng, d, nn, p, q, 4096)
            a = L.ints_to_limbs(vals, kw)
            m = L.int_to_limbs(nn, kw)[None, :]
            oo, so = oracle.modinv(mod_bits, a, m, 0)
            og = np.zeros_like(a); sg = np.full(len(vals), 9, np.uint8)
            t0 = time.time()
            ctx.modinv(mod_bits, len(vals), a, m, 0, og, sg)
            dt = time.time() - t0
            assert np.array_equal(so, sg) and np.array_equal(oo, og), (rd, mod_bits)
            total += len(vals)
            print(f"round {rd} modinv {mod_bits}: {len(vals)} items ok ({int((so == 1).sum())} without inverse), {dt * 1e3:.1f} ms incl. PCIe", flush=True)
    # ---- throughput with device-resident buffers (n = 2048)
    import torch
    p, q, n = H.fixture_key()
    nn = n * n
    for B in (4096, 16384):
        d = pm.Drbg(b"soak-inv-tp")
        vals = [d.below(nn) for _ in range(256)]
        a = torch.from_numpy(np.tile(L.ints_to_limbs(vals, 128), (B // 256, 1)).view(np.int32)).cuda()
        m = torch.from_numpy(L.int_to_limbs(nn, 128).view(np.int32)).cuda()
        out = torch.zeros_like(a); st = torch.zeros(B, dtype=torch.uint8, device="cuda")
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            ctx.modinv(4096, B, a, m, 0, out, st); ctx.synchronize()
            dt = time.time() - t0
        print(f"modinv 4096-bit, B={B}: {dt * 1e3:.1f} ms -> {B / dt:.0f} inverses/s", flush=True)
    # MulProof / CorrectMessageProof throughput, n = 2048, shared key, host buffers converted once
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_mul_and_message_proofs as T
    B = 512
    rows, arr = T.mul_cases(2048, [n], 64, b"soak-mul")
    dev = lambda x: torch.from_numpy(np.tile(x, (B // 64, 1)).view(np.int32)).cuda()
    ins = [dev(arr[k]) for k in T.MUL_IN]
    nd = torch.from_numpy(arr["n"][:1].view(np.int32)).cuda()
    f = torch.zeros((B, 64), dtype=torch.int32, device="cuda")
    z1, z2, e_d, e_db = (torch.zeros((B, 128), dtype=torch.int32, device="cuda") for _ in range(4))
    st = torch.zeros(B, dtype=torch.uint8, device="cuda")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        ctx.mul_proof_prove(2048, B, nd, 0, *ins, f, z1, z2, e_d, e_db, st); ctx.synchronize()
        tp = time.time() - t0
        t0 = time.time()
        ctx.mul_proof_verify(2048, B, nd, 0, ins[0], ins[1], ins[2], f, z1, z2, e_d, e_db, st); ctx.synchronize()
        tv = time.time() - t0
    print(f"MulProof n=2048 B={B}: prove {B / tp:.0f}/s, verify {B / tv:.0f}/s, all accepted: {bool((st == 1).all())}", flush=True)
    print("soak ok:", total, "inverses compared")


if __name__ == "__main__":
    main()
