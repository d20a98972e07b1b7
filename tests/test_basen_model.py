"""The base-n form of Paillier's arithmetic modulo n^2 (csrc/kernels_basen.hpp), stated in Python (tests/basen_model.py) and checked
against pow(): the algebra, the per-key constants, the bounds the ladder relies on — and, at lane level, the two things the kernels add
to the product of tests/test_lane_model.py: initial column values on the b side and the quotient digits taken from the a side, with the
64-bit column bound of the FAST product restated for the extra 58-bit value a b-side column starts with."""
import random

import pytest

from basen_model import BaseN, LB, B, MASK, W
from test_lane_model import to_limbs, from_limbs, fast_sn_limit


def odd_modulus(rnd, bits):
    return rnd.getrandbits(bits) | 1 | (1 << (bits - 1))


@pytest.mark.parametrize("bits,G", [(2048, 2), (2047, 2), (1200, 2), (4096, 4)])
def test_products_squarings_and_constants(bits, G):
    rnd = random.Random(bits * 7 + G)
    n = odd_modulus(rnd, bits)
    m = BaseN(n, G)
    nn = n * n
    Rinv = pow(m.R, -1, nn)
    assert m.value(m.one) == m.R % nn and m.value(m.RR) == m.R * m.R % nn
    assert 0 <= m.C3 < n
    for _ in range(8):
        x = (rnd.randrange(2 * m.Mt), rnd.randrange(4 * m.Mt))
        y = (rnd.randrange(2 * m.Mt), rnd.randrange(4 * m.Mt))
        z = m.mul(x, y)
        assert m.value(z) == m.value(x) * m.value(y) * Rinv % nn
        # the bounds are closed under the ladder's operations: a parts below 2 M~, b parts below 4 M~ (a sum of two products)
        assert z[0] < 2 * m.Mt and z[1] < 4 * m.Mt
        s = m.sqr(x)
        assert m.value(s) == m.value(x) ** 2 * Rinv % nn and s[0] < 2 * m.Mt and s[1] < 2 * m.Mt
    # into and out of the Montgomery domain, with any 2048-bit r (r >= n included) and the plain pair (1, m) on the way out
    r, msg = rnd.getrandbits(bits), rnd.randrange(n)
    x0 = m.to_mont(r)
    assert m.value(x0) == r * m.R % nn
    a0, bf = m.finish(m.sqr(x0), msg)
    assert 0 <= a0 < n and 0 <= bf < n and a0 + bf * n == r * r * (1 + msg * n) % nn


@pytest.mark.parametrize("bits", [512, 2048])
def test_enc_equals_pow(bits):
    """(1 + m n) r^n mod n^2 by square-and-multiply on pairs (512 bits on the 2048-bit geometry as well: the form needs n above half the
    capacity only for the constants' sake, k_setup_basen flags shorter keys)"""
    rnd = random.Random(bits)
    n = odd_modulus(rnd, bits)
    if bits < 1108:
        pytest.skip("below the length k_setup_basen accepts for this geometry (CAP / 2 + 64 bits)")
    m = BaseN(n, 2)
    for _ in range(2):
        r, msg = rnd.getrandbits(bits), rnd.randrange(n)
        assert m.enc(msg, r) == (1 + msg * n) * pow(r, n, n * n) % (n * n)


def sn_limit_basen(Wd=W):
    """kernels_basen.hpp COL_FAST_SN_LIMIT_BN"""
    return ((1 << 64) - 1 - (1 << 36) - ((1 << LB) + 16) * (Wd * (1 << LB) + 16) - (1 << (2 * LB)) - (1 << LB)) >> LB


def lane_product(N, G, A, Bv, init, stats):
    """bn_mul at lane level: montmul<G, ORUP, FAST> of test_lane_model.py from given initial columns; returns (result limbs, quotient digits)"""
    c = [[init[j * W + k] for k in range(W)] for j in range(G)]
    Q = []
    for s in range(G):
        for t in range(W):
            b = Bv[s * W + t]
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += A[j * W + k] * b
            q = c[0][t] & MASK
            Q.append(q)
            lo = [0] * G
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += N[j * W + k] * q
                    stats["maxcol"] = max(stats["maxcol"], c[j][(t + k) % W])
                v = c[j][t]
                lo[j] = v & MASK
                c[j][(t + 1) % W] += v >> LB
                stats["maxcol"] = max(stats["maxcol"], c[j][(t + 1) % W])
            assert lo[0] == 0
            for j in range(G):
                c[j][t] = lo[j + 1] if j + 1 < G else 0
    out, carries = [], []
    for j in range(G):
        cy, r = 0, []
        for k in range(W):
            v = c[j][k] + cy
            r.append(v & MASK)
            cy = v >> LB
        out.append(r)
        carries.append(cy)
    for j in range(1, G):
        out[j][0] += carries[j - 1]
    assert carries[G - 1] == 0
    return [v for r in out for v in r], Q


@pytest.mark.parametrize("bits,G", [(2048, 2), (4096, 4)])
def test_lane_level_b_side_takes_the_a_sides_digits(bits, G):
    rnd = random.Random(bits + 99)
    n = odd_modulus(rnd, bits)
    m = BaseN(n, G)
    L = m.L
    N = to_limbs(m.Mt, L)
    assert all(sum(N[j * W:(j + 1) * W]) <= sn_limit_basen() for j in range(G)), "a random key passes the digit-sum test"
    stats = {"maxcol": 0}
    zero = [0] * L
    for _ in range(3):
        x = (rnd.randrange(2 * m.Mt), rnd.randrange(4 * m.Mt))
        y = (rnd.randrange(2 * m.Mt), rnd.randrange(4 * m.Mt))
        # a side: the digits the array produces ARE the Q of the value-level model
        a_limbs, Q = lane_product(N, G, to_limbs(x[0], L), to_limbs(y[0], L), zero, stats)
        a_val, Qv = m.redc(x[0] * y[0])
        assert from_limbs(a_limbs) == a_val and from_limbs(Q) == Qv
        # b side: column i starts at C3_i + (2^29 - Q_i) n1
        c3 = to_limbs(m.C3, L)
        init = [c3[i] + (B - Q[i]) * m.n1 for i in range(L)]
        b1, _ = lane_product(N, G, to_limbs(x[0], L), to_limbs(y[1], L), init, stats)
        b2, _ = lane_product(N, G, to_limbs(x[1], L), to_limbs(y[0], L), zero, stats)
        want = m.mul(x, y)
        assert (a_val, from_limbs(b1) + from_limbs(b2)) == want
    assert stats["maxcol"] < (1 << 64)


def test_column_bound_with_the_initial_value():
    """worst-case operands (every limb at its maximum, plus the slack of almost-normalised operands) and a modulus operand exactly at
    COL_FAST_SN_LIMIT_BN: no column exceeds 64 bits although every column of the b side starts at up to 2^58 + 2^29"""
    lim = sn_limit_basen()
    assert lim < fast_sn_limit(W) and 26 * (1 << LB) < lim < 28 * (1 << LB)
    full = lim // MASK
    lane = [MASK] * full + [lim - full * MASK] + [0] * (W - full - 1)
    assert sum(lane) == lim
    # the analytic bound of bigint29.hpp "column capacity" with the b side's initial column value (< 2^58 + 2^29) added:
    assert ((1 << LB) + 16) * (W * (1 << LB) + 16) + MASK * lim + (1 << 36) + (1 << (2 * LB)) + (1 << LB) < (1 << 64)
    assert ((1 << LB) + 16) * (W * (1 << LB) + 16) + MASK * (lim + (1 << LB)) + (1 << 36) + (1 << (2 * LB)) + (1 << LB) >= (1 << 64) - (1 << 59)
