#!/usr/bin/env python3
"""Randomised soak of the L1 kernels against the oracle (run by hand on a GPU box: python tests/soak_gpu.py [rounds]).
Operands are drawn from adversarial families as well as uniformly: limbs of all ones (carry ripples in the exact
normalisation), values just below / above the modulus, moduli close to powers of two, sparse moduli, short moduli in
wide contexts, exponents with long runs of zeros or ones."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from helpers import pm, L
import oracle_lib


def family(d, bits, kind):
    top = (1 << bits) - 1
    if kind == 0: return d.bits(bits)
    if kind == 1: return top
    if kind == 2: return top ^ d.bits(bits // 8)
    if kind == 3: return (1 << (bits - 1)) | d.bits(64)
    if kind == 4: return d.bits(bits) & ~((1 << (bits // 2)) - 1) | ((1 << 29) - 1)
    if kind == 5: return d.bits(bits // 3)
    if kind == 6:  # runs of ones at 29-bit limb boundaries
        v = 0
        for i in range(0, bits, 29):
            if d.bytes(1)[0] & 1: v |= ((1 << 29) - 1) << i
        return v & top
    return d.bits(bits) | (top << (bits - 64)) & top


def run(ctx, oracle, tag=b"soak", rounds=3, deadline=None, log=print):
    """`rounds` rounds (or until time.monotonic() passes `deadline`) of inputs drawn from the DRBG seeded with `tag`; -> items checked"""
    import time
    total = 0
    for rd in range(rounds):
        for mod_bits, count, exp_bits in ((2048, 4096, 2048), (4096, 1024, 256), (4096, 512, 4096), (8192, 128, 512)):
            d = pm.Drbg(tag + b"-%d-%d-%d" % (rd, mod_bits, exp_bits))
            nl, el = mod_bits // 32, exp_bits // 32
            mods, bases, exps = [], [], []
            for i in range(count):
                m = family(d, mod_bits, i % 8) | 1
                if m < 3: m = 3
                mods.append(m)
                bases.append(family(d, mod_bits, (i // 8) % 8))
                exps.append(family(d, exp_bits, (i // 64) % 8))
            b, e, m = L.ints_to_limbs(bases, nl), L.ints_to_limbs(exps, el), L.ints_to_limbs(mods, nl)
            out = np.zeros_like(b)
            ctx.modexp(mod_bits, exp_bits, count, b, e, el, m, nl, out)
            ref = oracle.modexp(mod_bits, exp_bits, b, e, el, m, nl)
            bad = [i for i in range(count) if not np.array_equal(out[i], ref[i])]
            assert not bad, (tag, mod_bits, exp_bits, bad[:5])
            a2 = L.ints_to_limbs([family(d, mod_bits, (i // 3) % 8) for i in range(count)], nl)
            out2 = np.zeros_like(b)
            ctx.modmul(mod_bits, count, b, a2, m, nl, out2)
            assert np.array_equal(out2, oracle.modmul(mod_bits, b, a2, m, nl)), (tag, mod_bits)
            total += 2 * count
            # ONE exponent and ONE modulus for the whole call (the sliding-window ladder): every base family under a modulus of every family
            cs = min(count, 256)
            for fam in (rd % 8, (rd + 3) % 8):
                m1 = L.ints_to_limbs([max(3, family(d, mod_bits, fam) | 1)], nl)
                e1 = L.ints_to_limbs([family(d, exp_bits, (fam + rd) % 8)], el)
                o1 = np.zeros_like(b[:cs])
                ctx.modexp(mod_bits, exp_bits, cs, b[:cs], e1, 0, m1, 0, o1)
                r1 = oracle.modexp(mod_bits, exp_bits, b[:cs], np.repeat(e1, cs, axis=0), el, np.repeat(m1, cs, axis=0), nl)
                assert np.array_equal(o1, r1), (tag, "shared", mod_bits, exp_bits, fam)
                total += cs
            if deadline is not None and time.monotonic() > deadline:
                log("soak: time box reached in round", rd, "after", total, "items")
                return total
        # Paillier Enc with adversarial m, r under per-item keys, and under one shared key
        for n_bits, count in ((1024, 512), (2048, 512)):
            d = pm.Drbg(tag + b"-enc-%d-%d" % (rd, n_bits))
            kw = n_bits // 32
            ns = [family(d, n_bits, i % 8) | 1 | (1 << (n_bits - 1)) for i in range(count)]
            ms = [family(d, n_bits, (i // 8) % 8) for i in range(count)]
            rs = [family(d, n_bits, (i // 64) % 8) for i in range(count)]
            nl, m, r = (L.ints_to_limbs(v, kw) for v in (ns, ms, rs))
            out = np.zeros((count, 2 * kw), np.uint32)
            ctx.paillier_enc(n_bits, count, nl, kw, m, r, out)
            assert np.array_equal(out, oracle.paillier_enc(n_bits, nl, kw, m, r)), (tag, "enc", n_bits)
            n1 = np.ascontiguousarray(nl[rd % count:rd % count + 1])
            ctx.paillier_enc(n_bits, count, n1, 0, m, r, out)
            assert np.array_equal(out, oracle.paillier_enc(n_bits, np.repeat(n1, count, axis=0), kw, m, r)), (tag, "enc shared", n_bits)
            total += 2 * count
        log("round", rd, "ok,", total, "items checked so far")
        if deadline is not None and time.monotonic() > deadline:
            return total
    return total


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ctx = H.zkp.Context(0)
    oracle = oracle_lib.Oracle()
    oracle.set_threads(min(16, oracle.max_threads()))
    print("SOAK OK", run(ctx, oracle, b"soak", rounds, log=lambda *a: print(*a, flush=True)))


if __name__ == "__main__":
    main()
