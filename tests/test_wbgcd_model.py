"""The word-for-word Python model of csrc/kernels_gcd.hpp (tools/wbgcd_model.py): word-batched binary GCD / modular inverse with the
kernel's passes and int64 range assertions, against math.gcd / pow(x, -1, m) on random and structured operands."""
import math
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import wbgcd_model as W  # noqa: E402


@pytest.mark.parametrize("kw", [2, 8, 64, 128])
def test_inverse_and_gcd(kw):
    rnd = random.Random(1000 + kw)
    bits = 32 * kw
    stats = {"maxcof": 0}
    for trial in range(12 if kw > 8 else 120):
        m = rnd.getrandbits(bits) | 1 | (1 << (bits - 1)) if trial % 3 else max(3, rnd.getrandbits(rnd.randrange(3, bits)) | 1)
        y = [rnd.randrange(1, m), 1, m - 1, 1 << rnd.randrange(0, m.bit_length() - 1), max(1, rnd.randrange(1, m) >> rnd.randrange(0, bits)), max(1, m // 2)][trial % 6]
        if trial % 5 == 0 and kw > 2:                     # a common factor: no inverse
            p = rnd.getrandbits(bits // 3) | 1
            m = (p * (rnd.getrandbits(bits - bits // 3 - 1) | 1)) | 1
            if m % p == 0 and m // p > 1:
                y = p * rnd.randrange(1, m // p)
        g, inv = W.wbgcd(y, m, kw, True, stats)
        assert g == math.gcd(y, m)
        if g == 1:
            assert inv == pow(y, -1, m)
        assert W.wbgcd(y, m, kw, False)[0] == g
    assert stats["maxcof"] <= 1 and stats.get("final_passes", 0) <= 2   # the cofactors stay within a small multiple of m


def test_cofactor_slightly_above_the_modulus_in_magnitude():
    """a = M - r for small even r: the balanced correction leaves v = -1.07 M (advisor's vector, round 2).  A single conditional
    add of M returned a negative number with status OK; the finalisation reduces until 0 <= v < M."""
    M = 4400505965657808285
    stats = {"maxcof": 0}
    g, inv = W.wbgcd(M - 14, M, 8, True, stats)
    assert g == 1 and inv == pow(M - 14, -1, M) and stats["maxcof"] == 1 and stats["final_passes"] == 2
    rnd = random.Random(99)
    hits = 0
    for bits in (123, 153, 2041, 8192):   # bit lengths congruent to 1..6 mod 30 are where it shows
        kw = ((bits + 31) // 32 + 7) // 8 * 8
        for _ in range(40 if bits < 1000 else 6):
            m = rnd.getrandbits(bits) | 1 | (1 << (bits - 1))
            for r in (14, 30, 42, 46, 62):
                if math.gcd(m - r, m) != 1:
                    continue
                st = {"maxcof": 0}
                g, inv = W.wbgcd(m - r, m, kw, True, st)
                assert g == 1 and inv == pow(m - r, -1, m)
                hits += st["maxcof"] > 0
    assert hits > 0                                      # the case is exercised, not just survived


def test_gcd_of_values_above_the_modulus_and_zero_low_words():
    """the DLog coprimality tests hand over any kw-word value, also ones above N and ones with zero low words"""
    rnd = random.Random(7)
    kw = 64
    N = rnd.getrandbits(2048) | 1 | (1 << 2047)
    for x in (N + 2, (1 << 2048) - 1, 3 << 1000, N << 0, (N >> 1) << 1, 1 << 2047):
        x &= (1 << 2048) - 1
        if x == 0:
            continue
        assert W.wbgcd(x, N, kw, False)[0] == math.gcd(x, N)
