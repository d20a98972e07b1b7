"""CPU model of the wave-group Montgomery multiplication of csrc/bigint29.hpp (word-level CIOS on G lanes x W
limbs of 29 bits, circular column window, optional Orup multiple), checked against Python integers.  It documents
the invariants the kernel relies on: no 64-bit column overflow, lane 0's bottom limb always zero (so the DPP
pass-down needs no masking between groups), result < 2M without any conditional subtraction, <= M when B == 1."""
import random

import pytest

B = 29
MASK = (1 << B) - 1


def to_limbs(x, n):
    return [(x >> (B * i)) & MASK for i in range(n)]


def from_limbs(l):
    return sum(v << (B * i) for i, v in enumerate(l))


def montmul(N, n1, G, W, A, Bv, orup, stats):
    """mirror of montmul<G, ORUP>: N = limbs of M (or of M~ = M*n1 when orup), n1 = -M^-1 mod 2^29"""
    c = [[0] * W for _ in range(G)]
    for s in range(G):
        for t in range(W):
            b = Bv[s * W + t]
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += A[j * W + k] * b
            c0 = c[0][t] & 0xFFFFFFFF
            q = (c0 if orup else (c0 * n1) & 0xFFFFFFFF) & MASK
            lo = [0] * G
            for j in range(G):
                for k in range(W):
                    c[j][(t + k) % W] += N[j * W + k] * q
                    stats["maxcol"] = max(stats["maxcol"], c[j][(t + k) % W])
                v = c[j][t]
                lo[j] = v & MASK
                c[j][(t + 1) % W] += v >> B
            assert lo[0] == 0                      # the value lane G-1 of the previous group would receive
            for j in range(G):
                c[j][t] = lo[j + 1] if j + 1 < G else 0
    out, carries = [], []
    for j in range(G):
        cy, r = 0, []
        for k in range(W):
            v = c[j][k] + cy
            r.append(v & MASK)
            cy = v >> B
        out.append(r)
        carries.append(cy)
    for j in range(1, G):
        out[j][0] += carries[j - 1]
        assert out[j][0] < (1 << B) + 64
    assert carries[G - 1] == 0
    return [v for r in out for v in r]


@pytest.mark.parametrize("bits,G,W", [(2048, 4, 18), (4096, 8, 18), (2048, 8, 9), (4096, 16, 9), (300, 4, 18)])
@pytest.mark.parametrize("orup", [False, True])
def test_word_level_cios_model(bits, G, W, orup):
    rnd = random.Random(bits * 31 + G + W + orup)
    M = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
    L = G * W
    R = 1 << (B * L)
    n1 = (-pow(M, -1, 1 << B)) % (1 << B)
    Mt = M * n1
    assert Mt % (1 << B) == MASK and 4 * Mt < R      # Orup multiple fits with headroom
    N = to_limbs(Mt if orup else M, L)
    bound = 2 * (Mt if orup else M)
    Rinv = pow(R, -1, M)
    stats = {"maxcol": 0}
    for _ in range(2):
        a, b = rnd.randrange(bound), rnd.randrange(bound)
        r = montmul(N, n1, G, W, to_limbs(a, L), to_limbs(b, L), orup, stats)
        v = from_limbs(r)
        assert v % M == a * b * Rinv % M and v < bound
        r2 = montmul(N, n1, G, W, r, r, orup, stats)           # feed the almost-normalised output straight back
        assert from_limbs(r2) % M == v * v * Rinv % M
    if not orup:
        one = montmul(N, n1, G, W, to_limbs(rnd.randrange(2 * M), L), to_limbs(1, L), orup, stats)
        assert from_limbs(one) <= M                            # montmul(x, 1) <= M: one equality test canonicalises
    assert stats["maxcol"] < (1 << 64)
