"""The base-n form of Paillier's arithmetic modulo n^2 (csrc/kernels_basen.hpp) on the GPU against tests/basen_model.py: constants and
single operations limb for limb through the diagnostics entry point, then whole Enc calls against Python's pow() through the ordinary
entry points (the n^2-sized kernels answer the same calls in tests/test_gpu_l1.py under the `w36-n2` context)."""
import random

import numpy as np
import pytest

import helpers as H
from basen_model import BaseN, LB, MASK

zkp = H.zkp
pytestmark = pytest.mark.gpu


def limbs(x, L):
    return np.array([(x >> (LB * i)) & MASK for i in range(L)], np.uint32)


def value(arr):
    return sum(int(v) << (LB * i) for i, v in enumerate(arr))


def words(x, n):
    return np.array([(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)], np.uint32)


def odd_modulus(rnd, bits):
    return rnd.getrandbits(bits) | 1 | (1 << (bits - 1))


@pytest.fixture(scope="module")
def ctx():
    """a context of its own: the batches here are small, so every Paillier launch is told to take the base-n form (left to itself the
    library keeps them on the n^2-sized kernels: launch_basen's routing rule, tests/test_gpu_routing.py)"""
    c = zkp.Context(0)
    c.set_enc_form("basen")
    yield c
    c.close()


@pytest.mark.parametrize("n_bits", [2048, 4096])
def test_constants_and_single_operations_match_the_model(ctx, n_bits):
    rnd = random.Random(n_bits)
    G = n_bits // 1024
    for trial in range(3):
        n = odd_modulus(rnd, n_bits - (trial == 2))           # a key one bit short of the field as well
        m = BaseN(n, G)
        L = m.L
        nw = words(n, n_bits // 32)
        out = ctx.diag_basen(n_bits, nw, 3)
        assert int(out[4 * L + 1]) == 1, "the key should qualify for the base-n form"
        assert int(out[4 * L]) == m.n1
        assert value(out[3 * L:4 * L]) == m.Mt
        assert value(out[0:L]) % n == m.C3 % n and value(out[0:L]) <= n
        assert (value(out[L:2 * L]) + value(out[2 * L:3 * L]) * n) % (n * n) == (m.R * m.R) % (n * n)
        rr = (value(out[L:2 * L]), value(out[2 * L:3 * L]))
        # the kernels' RR need not be the canonical pair: every later check uses the value the device holds
        m.RR = rr
        # to the Montgomery domain
        r = rnd.getrandbits(n_bits)
        o = ctx.diag_basen(n_bits, nw, 0, xa=limbs(r, L))
        # C3 on the device may be n - val with val == 0 -> n; the model's b sides must use the same constant
        m.C3 = value(out[0:L])
        want = m.mul((r, 0), rr)
        assert (value(o[:L]), value(o[L:2 * L])) == want
        # squaring and product of arbitrary pairs below the bounds the ladder keeps
        for _ in range(3):
            x = (rnd.randrange(2 * m.Mt), rnd.randrange(4 * m.Mt))
            y = (rnd.randrange(2 * m.Mt), rnd.randrange(4 * m.Mt))
            o = ctx.diag_basen(n_bits, nw, 2, xa=limbs(x[0], L), xb=limbs(x[1], L))
            assert (value(o[:L]), value(o[L:2 * L])) == m.sqr(x)
            o = ctx.diag_basen(n_bits, nw, 1, xa=limbs(x[0], L), xb=limbs(x[1], L), ya=limbs(y[0], L), yb=limbs(y[1], L))
            # the kernel multiplies the STAGED x by the resident y: mul(y, x) in the model's argument order
            assert (value(o[:L]), value(o[L:2 * L])) == m.mul(y, x)


def test_keys_outside_the_form_are_flagged(ctx):
    rnd = random.Random(7)
    L = 72
    short = odd_modulus(rnd, 1000)                              # far too short for the b parts
    out = ctx.diag_basen(2048, words(short, 64), 3)
    assert int(out[4 * L + 1]) == 0
    even = odd_modulus(rnd, 2048) - 1
    out = ctx.diag_basen(2048, words(even, 64), 3)
    assert int(out[4 * L + 1]) == 0


@pytest.mark.parametrize("n_bits", [2048, 4096])
def test_enc_batch_equals_python_and_the_n2_sized_kernels(ctx, n_bits):
    rnd = random.Random(n_bits + 1)
    kw = n_bits // 32
    n = odd_modulus(rnd, n_bits)
    nn = n * n
    count = 70                                                  # more than one wavefront of groups, ragged
    ms = [rnd.randrange(n) for _ in range(count)]
    rs = [rnd.getrandbits(n_bits) for _ in range(count)]        # r >= n included: (r + k n)^n == r^n (mod n^2)
    ms[0], rs[0] = 0, 1
    ms[1], rs[1] = n - 1, n - 1
    ms[2], rs[2] = (1 << n_bits) - 1, (1 << n_bits) - 1          # both above n: only m mod n and r mod n matter
    ms[3], rs[3] = 5, 0                                            # r = 0 and r = n: Enc = 0
    ms[4], rs[4] = 7, n
    ms[5], rs[5] = n, n + 1                                        # m = n: the factor (1 + m n) is 1 modulo n^2
    nw = words(n, kw)
    mw = np.stack([words(v, kw) for v in ms])
    rw = np.stack([words(v, kw) for v in rs])
    out = np.zeros((count, 2 * kw), np.uint32)
    ctx.set_geometry(zkp.load().zkp_build_limbs_per_lane())    # the throughput engine (a 70-item call would go to the latency engine)
    ctx.paillier_enc(n_bits, count, nw, 0, mw, rw, out)
    for i in range(count):
        got = sum(int(w) << (32 * j) for j, w in enumerate(out[i]))
        assert got == (1 + ms[i] * n) * pow(rs[i], n, nn) % nn, i
    # Enc-and-compare: right and wrong expected values, and products of two ciphertexts as the expected value
    exp = out.copy()
    exp[6, 5] ^= 1
    ok = np.full(count, 9, np.uint8)
    ctx.paillier_enc_check(n_bits, count, nw, 0, mw, rw, None, None, exp, ok)
    assert list(ok) == [0 if i == 6 else 1 for i in range(count)]
    # expected = a * b mod n^2 (the Mask rows of RangeProofNi::verify): a = Enc(m, r) / b for an invertible b
    import math
    bs = []
    while len(bs) < count:
        v = rnd.randrange(2, nn)
        if math.gcd(v, n) == 1:
            bs.append(v)
    a_ = [(sum(int(w) << (32 * j) for j, w in enumerate(out[i])) * pow(bs[i], -1, nn)) % nn for i in range(count)]
    a_[8] = (a_[8] + 1) % nn
    aw = np.stack([words(v, 2 * kw) for v in a_])
    bw = np.stack([words(v, 2 * kw) for v in bs])
    ok = np.full(count, 9, np.uint8)
    ctx.paillier_enc_check(n_bits, count, nw, 0, mw, rw, aw, bw, None, ok)
    assert list(ok) == [0 if i == 8 else 1 for i in range(count)]
    ctx.set_geometry(0)


@pytest.mark.parametrize("n_bits", [2048, 4096])
def test_enc_batch_with_per_item_keys_equals_python(ctx, n_bits):
    """n_stride != 0: every item under its own key (fixed-window ladder over the item's n, constants per key, C3 from global memory)"""
    rnd = random.Random(n_bits + 2)
    kw = n_bits // 32
    count = 40 if n_bits == 2048 else 20
    ns = [odd_modulus(rnd, n_bits - (i % 3 == 2)) for i in range(count)]       # some keys one bit short of the field
    ms = [rnd.randrange(n) for n in ns]
    rs = [rnd.getrandbits(n_bits) for _ in range(count)]
    ms[0], rs[0] = 0, 1
    nw = np.stack([words(v, kw) for v in ns])
    mw = np.stack([words(v, kw) for v in ms])
    rw = np.stack([words(v, kw) for v in rs])
    out = np.zeros((count, 2 * kw), np.uint32)
    ctx.set_geometry(zkp.load().zkp_build_limbs_per_lane())
    ctx.paillier_enc(n_bits, count, nw, kw, mw, rw, out)
    lanes, ok = ctx.diag_basen_last()
    assert lanes == n_bits // 1024 and ok, "the per-key launch should have run in base-n form"
    for i in range(count):
        n = ns[i]
        got = sum(int(w) << (32 * j) for j, w in enumerate(out[i]))
        assert got == (1 + ms[i] * n) * pow(rs[i], n, n * n) % (n * n), i
    # one key that does not qualify (even) sends the WHOLE launch to the n^2-sized kernels: same answers for the others, zeros for it
    ns2 = list(ns)
    ns2[3] -= 1
    nw2 = np.stack([words(v, kw) for v in ns2])
    out2 = np.zeros((count, 2 * kw), np.uint32)
    try:
        ctx.paillier_enc(n_bits, count, nw2, kw, mw, rw, out2)
    except zkp.ZkpError:
        pass                                                   # (the entry point reports the even modulus; the outputs of the others are written)
    lanes, ok = ctx.diag_basen_last()
    assert lanes == n_bits // 1024 and not ok
    for i in range(count):
        if i != 3:
            assert np.array_equal(out2[i], out[i]), i
    ctx.set_geometry(0)
