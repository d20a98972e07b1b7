#!/usr/bin/env python3
"""Randomised soak of the latency engine's right-to-left Enc ladders (csrc/kernels_basen_r2l.hpp) against Python's pow(): both lane
geometries — five wavefronts of 36 lanes per Enc (k_enc_basen_r2l5), one wavefront of five 12-lane groups (k_enc_basen_r2l<6>) — under
moduli and operands from adversarial families (limbs of all ones, moduli close to powers of two, sparse and short moduli, r >= n, m >= n).
Run by hand on a GPU box: python tests/soak_gpu_r2l.py [seconds] [seed]"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H

zkp = H.zkp
N_BITS, KW = 2048, 64


def words(v, n):
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)], np.uint32)


def value(rnd, bits, kind):
    top = (1 << bits) - 1
    if kind == 0: return rnd.getrandbits(bits)
    if kind == 1: return top
    if kind == 2: return top ^ rnd.getrandbits(bits // 8)
    if kind == 3: return (1 << (bits - 1)) | rnd.getrandbits(64)
    if kind == 4: return (rnd.getrandbits(bits) & ~((1 << (bits // 2)) - 1)) | ((1 << 29) - 1)
    if kind == 5: return rnd.getrandbits(bits // 3)
    if kind == 6:                       # runs of ones at 29-bit limb boundaries
        v = 0
        for i in range(0, bits, 29):
            if rnd.getrandbits(1): v |= ((1 << 29) - 1) << i
        return v & top
    return rnd.getrandbits(bits) | ((top << (bits - 64)) & top)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    print(f"seed {seed}", flush=True)
    rnd = random.Random(seed)
    c = zkp.Context(0)
    c.set_geometry(9); c.set_r2l(2)
    t_end = time.monotonic() + seconds
    trials = items = on_ladder = 0
    while time.monotonic() < t_end:
        bits = rnd.choice((2048, 2048, 2047, 2040, 1800, 1500, 1200, 1100))
        n = value(rnd, bits, rnd.randrange(8)) | 1 | (1 << (bits - 1))
        nn = n * n
        count = rnd.choice((1, 2, 3, 5, 17, 64, 100, 256, 300))
        ms = [value(rnd, N_BITS, rnd.randrange(8)) % ((1 << N_BITS)) for _ in range(count)]
        rs = [value(rnd, N_BITS, rnd.randrange(8)) for _ in range(count)]
        for i in range(count):
            if rnd.randrange(4):
                ms[i] %= n; rs[i] %= n              # the honest case three times out of four
        nw = words(n, KW)
        mw = np.stack([words(v, KW) for v in ms]); rw = np.stack([words(v, KW) for v in rs])
        outs = {}
        for lanes in (36, 12):
            c.set_r2l_lanes(lanes)
            out = np.zeros((count, 2 * KW), np.uint32)
            c.paillier_enc(N_BITS, count, nw, 0, mw, rw, out)
            outs[lanes] = out
            on_ladder += int(c.r2l_lanes_last() == lanes)
        assert np.array_equal(outs[36], outs[12]), (seed, trials, "the two geometries differ")
        for i in range(count):
            got = sum(int(w) << (32 * j) for j, w in enumerate(outs[36][i]))
            assert got == (1 + ms[i] * n) * pow(rs[i], n, nn) % nn, (seed, trials, i, hex(n))
        trials += 1; items += count
    c.close()
    print(f"ok: {trials} keys, {items} Enc on each geometry ({on_ladder} of {2 * trials} launches ran on the ladder they were pinned to), all equal to pow()", flush=True)


if __name__ == "__main__":
    main()
