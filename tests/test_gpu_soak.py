"""Randomised and boundary coverage inside the -m gpu suite (round-2 verdict: the soaks ran by hand only).

* test_randomised_soak_*: the generators of tests/soak_gpu.py / soak_gpu_proofs.py under a time box, seeded per run.  The seed is
  printed and can be pinned: ZKP_SOAK_SEED=<int>; ZKP_SOAK_SECONDS sets the time box (default 25 s per soak).
* test_fast_product_limit_keys: moduli whose Orup multiple sits exactly at / one below / one above COL_FAST_SN_LIMIT in one lane —
  the switch between the FAST and the SAFE product (bigint29.hpp "column capacity", k_setup's digit-sum test) — with operands at
  their limb maxima, per-item and shared exponents (the latter: the squaring path of the sliding-window ladder).
* test_double_digit_headroom_keys: 2048-bit-context moduli of bl + 61 == capacity and +-1 bits (k_setup: mt2_ok, latency engine)."""
import os
import time

import numpy as np
import pytest

import helpers as H  # noqa: F401  (puts the repo root and tests/ on sys.path)
from helpers import pm, L
import soak_gpu
import soak_gpu_proofs

pytestmark = pytest.mark.gpu

LB = 29
MASK = (1 << LB) - 1


def _seed():
    s = os.environ.get("ZKP_SOAK_SEED")
    return int(s) if s else int(time.time() * 1000) & 0x7FFFFFFF


def _box():
    return float(os.environ.get("ZKP_SOAK_SECONDS", "12"))      # (the driver's GPU run is on a budget: tests/test_gpu_suite_budget.py; the stand-alone soaks tests/soak_gpu*.py run as long as asked)


def test_randomised_soak_l1(zkp, oracle):
    seed = _seed()
    print(f"\nZKP_SOAK_SEED={seed} (L1 soak; rerun with this value to reproduce)")
    ctx = zkp.Context(0)
    try:
        oracle.set_threads(min(16, oracle.max_threads()))
        total = 0
        for geom, form in ((36, "basen"), (36, "n2"), (9, "auto")):     # both Enc forms of the throughput engine, then the latency engine
            ctx.set_geometry(geom)
            ctx.set_enc_form(form)
            total += soak_gpu.run(ctx, oracle, b"soak-%d-%d-%s" % (seed, geom, form.encode()), rounds=1000, deadline=time.monotonic() + _box() / 3,
                                  log=lambda *a: print(*a, flush=True))
        assert total > 0
        print("L1 soak items:", total)
    finally:
        ctx.close()


def test_randomised_soak_proofs(zkp, oracle):
    seed = _seed()
    print(f"\nZKP_SOAK_SEED={seed} (proof soak; rerun with this value to reproduce)")
    ctx = zkp.Context(0)
    try:
        oracle.set_threads(min(16, oracle.max_threads()))
        total = soak_gpu_proofs.run(ctx, oracle, seed, rounds=1000, deadline=time.monotonic() + _box(), log=lambda *a: print(*a, flush=True))
        assert total > 0
        print("proofs soaked:", total)
    finally:
        ctx.close()


def fast_sn_limit(W=36):
    return ((1 << 64) - 1 - (1 << 36) - ((1 << LB) + 16) * (W * (1 << LB) + 16)) >> LB


def limit_modulus(mod_bits, lane, delta, d, W=36):
    """odd modulus M == -1 (mod 2^29) (so n' = 1 and the Orup multiple IS M) whose 29-bit limbs of lane `lane` sum to
    COL_FAST_SN_LIMIT + delta; the other lanes are random (mean 0.5)."""
    nlimbs = (mod_bits + LB - 1) // LB
    G = (nlimbs + W - 1) // W
    limbs = [d.bits(LB) for _ in range(G * W)]
    top_bits = mod_bits - (nlimbs - 1) * LB
    for i in range(nlimbs, G * W):
        limbs[i] = 0
    limbs[nlimbs - 1] = (d.bits(top_bits) | (1 << (top_bits - 1))) & ((1 << top_bits) - 1)
    want = fast_sn_limit(W) + delta
    lo, hi = lane * W, min((lane + 1) * W, nlimbs - 1)      # (leave the top limb of the integer alone)
    fixed = sum(limbs[hi:(lane + 1) * W])
    if lane == 0:
        limbs[0] = MASK
        fixed += MASK
        lo = 1
    want -= fixed
    n = hi - lo
    assert 0 <= want <= n * MASK, "target digit sum out of reach for this lane"
    full, rest = divmod(want, MASK)
    vals = [MASK] * full + ([rest] if full < n else []) + [0] * (n - full - 1)
    assert len(vals) == n and sum(vals) == want
    limbs[lo:hi] = vals
    limbs[0] = MASK
    if lane != 0:
        pass
    M = sum(v << (LB * i) for i, v in enumerate(limbs))
    assert M & MASK == MASK and M.bit_length() <= mod_bits
    assert sum((M >> (LB * i)) & MASK for i in range(lane * W, (lane + 1) * W)) == fast_sn_limit(W) + delta
    return M


@pytest.mark.parametrize("mod_bits", [2048, 4096])
def test_fast_product_limit_keys(ctx, oracle, mod_bits):
    d = pm.Drbg(b"limit-keys-%d" % mod_bits)
    nl = mod_bits // 32
    G = {2048: 2, 4096: 4}[mod_bits]
    mods = []
    for lane in range(G):
        for delta in (-1, 0, 1, 2 ** 20, -(2 ** 20)):
            mods.append(limit_modulus(mod_bits, lane, delta, d))
    count = len(mods)
    big = (1 << mod_bits) - 1
    for bases in ([big] * count, [m - 1 for m in mods], [d.bits(mod_bits) for _ in mods]):
        exps = [d.bits(mod_bits) | (1 << (mod_bits - 1)) for _ in mods]
        b, e, m = (L.ints_to_limbs(v, nl) for v in (bases, exps, mods))
        out = np.zeros_like(b)
        ctx.modexp(mod_bits, mod_bits, count, b, e, nl, m, nl, out)                     # per-item exponents: fixed windows
        assert np.array_equal(out, oracle.modexp(mod_bits, mod_bits, b, e, nl, m, nl))
        for i in range(count):                                                         # one modulus, one exponent: sliding windows + squarings
            bb = np.ascontiguousarray(np.repeat(b[i:i + 1], 5, axis=0)); bb[1:] = L.ints_to_limbs([big, 1, mods[i] - 1, d.below(mods[i])], nl)
            o = np.zeros_like(bb)
            ctx.modexp(mod_bits, mod_bits, 5, bb, e[i:i + 1], 0, m[i:i + 1], 0, o)
            assert np.array_equal(o, oracle.modexp(mod_bits, mod_bits, bb, e[i:i + 1], 0, m[i:i + 1], 0)), (i, "shared")


def test_double_digit_headroom_keys(ctx, oracle):
    """latency engine, 2048-bit context: capacity 8 x 9 x 29 = 2088 bits; the 58-bit Orup multiple needs bl + 61 <= 2088"""
    d = pm.Drbg(b"mt2-headroom")
    nl = 64
    mods, bases, exps = [], [], []
    for bl in (2025, 2026, 2027, 2028, 2029, 2048):
        for _ in range(3):
            m = d.bits(bl) | 1 | (1 << (bl - 1))
            mods.append(m); bases.append(d.below(m)); exps.append(d.bits(2048))
        mods.append((1 << bl) - 1); bases.append((1 << bl) - 2); exps.append((1 << 2048) - 1)
    b, e, m = (L.ints_to_limbs(v, nl) for v in (bases, exps, mods))
    out = np.zeros_like(b)
    ctx.modexp(2048, 2048, len(mods), b, e, nl, m, nl, out)
    assert np.array_equal(out, oracle.modexp(2048, 2048, b, e, nl, m, nl))
    # NiCorrectKeyProof's shape on the same moduli (sigma^n mod n, exponent = the modulus): ONE key per call -> shared exponent
    for i in (0, 4, 8, 12):
        o = np.zeros_like(b[:4])
        ctx.modexp(2048, 2048, 4, b[:4], m[i:i + 1], 0, m[i:i + 1], 0, o)
        assert np.array_equal(o, oracle.modexp(2048, 2048, b[:4], m[i:i + 1], 0, m[i:i + 1], 0))
