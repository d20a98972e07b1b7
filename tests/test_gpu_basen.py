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


@pytest.fixture(scope="module", params=[36, 18, 9], ids=["w36", "w18", "w9"])
def ctx(request):
    """a context of its own per engine — the form exists at 36 limbs per lane (2 / 4 lanes per n-sized integer, the kernels of the large
    batches) and at 9 (8 / 16 lanes, the latency engine: the kernels of the mid-size batches).  The batches here are small, so every
    Paillier launch is told to take the form (left to itself the library keeps them on the n^2-sized kernels: launch_basen's routing
    rule, tests/test_gpu_routing.py)"""
    c = zkp.Context(0)
    c.set_geometry(request.param)
    c.set_enc_form("basen")
    c.test_geometry = request.param
    yield c
    c.close()


def lanes_per_integer(ctx, n_bits):
    return (72 // ctx.test_geometry) * (n_bits // 2048)


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
    count = 70 if n_bits == 2048 else 24                        # more than one wavefront of groups, ragged (Python's pow on 8192-bit operands is the test's time: 0.3 s each)
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
    ctx.paillier_enc(n_bits, count, nw, 0, mw, rw, out)
    lanes, ok = ctx.diag_basen_last()
    assert lanes == lanes_per_integer(ctx, n_bits) and ok, "the launch should have run in base-n form"
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


def test_enc_under_keys_with_long_runs_of_equal_bits(ctx):
    """exponents the window script treats specially: runs of more than 31 squarings (the compact script of the assembler engine counts a
    run in one byte, 1 .. 31, and takes several for a longer one — kernels_modexp.hpp k_sliding_schedule), and runs of ones (a window
    multiplication after every window).  Shared key, 40 items each, against Python"""
    rnd = random.Random(99)
    n_bits, kw = 2048, 64
    keys = [(1 << 2047) + (1 << 1000) + 1,                                   # two runs of ~1000 zero bits
            (1 << 2048) - (1 << 900) - 1 - (1 << 37),                         # ones almost everywhere
            (1 << 2047) | (rnd.getrandbits(300) << 1500) | rnd.getrandbits(200) | 1]     # islands of random bits in zeros
    for n in keys:
        assert n & 1 and n.bit_length() == n_bits
        nn = n * n
        count = 40
        ms = [rnd.randrange(n) for _ in range(count)]
        rs = [rnd.randrange(n) for _ in range(count)]
        nw = words(n, kw)
        mw = np.stack([words(v, kw) for v in ms]); rw = np.stack([words(v, kw) for v in rs])
        out = np.zeros((count, 2 * kw), np.uint32)
        ctx.paillier_enc(n_bits, count, nw, 0, mw, rw, out)
        lanes, ok = ctx.diag_basen_last()
        for i in range(count):
            got = sum(int(w) << (32 * j) for j, w in enumerate(out[i]))
            assert got == (1 + ms[i] * n) * pow(rs[i], n, nn) % nn, (hex(n)[:20], i, lanes, ok)


@pytest.mark.parametrize("n_bits", [2048, 4096])
def test_enc_batch_with_per_item_keys_equals_python(ctx, n_bits):
    """n_stride != 0: every item under its own key (fixed-window ladder over the item's n, constants per key, C3 from global memory)"""
    rnd = random.Random(n_bits + 2)
    kw = n_bits // 32
    count = 40 if n_bits == 2048 else 12
    ns = [odd_modulus(rnd, n_bits - (i % 3 == 2)) for i in range(count)]       # some keys one bit short of the field
    ms = [rnd.randrange(n) for n in ns]
    rs = [rnd.getrandbits(n_bits) for _ in range(count)]
    ms[0], rs[0] = 0, 1
    nw = np.stack([words(v, kw) for v in ns])
    mw = np.stack([words(v, kw) for v in ms])
    rw = np.stack([words(v, kw) for v in rs])
    out = np.zeros((count, 2 * kw), np.uint32)
    ctx.paillier_enc(n_bits, count, nw, kw, mw, rw, out)
    lanes, ok = ctx.diag_basen_last()
    assert lanes == lanes_per_integer(ctx, n_bits) and ok, "the per-key launch should have run in base-n form"
    for i in range(count):
        n = ns[i]
        got = sum(int(w) << (32 * j) for j, w in enumerate(out[i]))
        assert got == (1 + ms[i] * n) * pow(rs[i], n, n * n) % (n * n), i
    # keys the form does not take are PARTITIONED off, item by item (round 5; round 4 sent the whole launch to the n^2-sized kernels):
    # a short key (1000 bits in the 2048-bit context: no room for the b parts) — its item comes back right from the n^2-sized launch behind
    ns2 = list(ns)
    ns2[5] = odd_modulus(rnd, n_bits // 2 - 24)
    ms2 = list(ms); ms2[5] = ms[5] % ns2[5]
    out2 = np.zeros((count, 2 * kw), np.uint32)
    ctx.paillier_enc(n_bits, count, np.stack([words(v, kw) for v in ns2]), kw, np.stack([words(v, kw) for v in ms2]), rw, out2)
    lanes, ok = ctx.diag_basen_last()
    assert lanes == lanes_per_integer(ctx, n_bits) and not ok          # (the batch-wide flag: not every key qualified)
    for i in range(count):
        n = ns2[i]
        got = sum(int(w) << (32 * j) for j, w in enumerate(out2[i]))
        assert got == (1 + ms2[i] * n) * pow(rs[i], n, n * n) % (n * n), i
    # an even key: the entry point reports it (no Montgomery form exists), the other items are done and equal to the clean batch's
    ns3 = list(ns)
    ns3[3] -= 1
    out3 = np.zeros((count, 2 * kw), np.uint32)
    try:
        ctx.paillier_enc(n_bits, count, np.stack([words(v, kw) for v in ns3]), kw, mw, rw, out3)
    except zkp.ZkpError:
        pass
    for i in range(count):
        if i != 3:
            assert np.array_equal(out3[i], out[i]), i


def test_range_proofs_under_per_proof_keys_with_keys_outside_the_form(oracle):
    """RangeProofNi prove + verify of a batch with per-proof keys in which two keys do not qualify for the base-n form (short ones): the
    base-n launch takes the other proofs' rows, the n^2-sized launch behind it the rows of those two — transcripts and verdicts against the
    oracle (range_proof.rs:270-348: every proof under its own ek)"""
    import importlib
    synth = importlib.import_module("zk-paillier_amd.synth")
    n_bits, B = 2048, 24
    keys = synth.distinct_keys_2048(B)
    keys[4] = H.test_key(1000, tag=1)[2]
    keys[17] = H.test_key(1024, tag=2)[2]
    cases = H.build_range_case(b"partition", keys, n_bits, B, shared=False)
    oracle.set_threads(min(16, oracle.max_threads()))
    pb_o, wt = H.fill_batch(cases, n_bits, False, oracle)
    oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
    c = zkp.Context(0)
    try:
        for geometry in (36, 18, 9):
            c.set_geometry(geometry)
            c.set_enc_form("basen")
            pb = zkp.RangeBatch(n_bits, B, 128, shared_key=False)
            pb.n[:] = pb_o.n; pb.range[:] = pb_o.range; pb.ciphertext[:] = pb_o.ciphertext
            c.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=False)
            lanes, ok = c.diag_basen_last()
            assert lanes == (72 // geometry) and not ok, (geometry, lanes, ok)
            for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
                assert np.array_equal(getattr(pb_o, f), getattr(pb, f)), (geometry, f)
            pb.resp_r1[4, 9, 0] ^= 1; pb.resp_r1[5, 9, 0] ^= 1; pb.c1[17, 100, 3] ^= 2
            vo = np.full(B, 9, np.uint8); vg = np.full(B, 9, np.uint8)
            oracle.range_ni_verify(pb.struct(), vo)
            c.range_ni_verify(pb.struct(), vg, device=False)
            assert list(vg) == list(vo) and list(vo) == [0 if b in (4, 5, 17) else 1 for b in range(B)]
    finally:
        c.close()


@pytest.mark.parametrize("lanes", [12, 8, 36], ids=["12-lanes-x-6-limbs", "8-lanes-x-9-limbs", "five-wavefronts-of-36-lanes-x-2-limbs"])
def test_one_enc_per_wavefront_ladder_of_the_latency_engine(lanes):
    """csrc/kernels_basen_r2l.hpp: the base-n exponentiation as a right-to-left ladder pipelined over five lane groups, the latency engine's
    kernel for calls of a few proofs under one 2048-bit key (tests/test_basen_r2l_model.py states the pipeline on values): Enc against
    Python's pow() with the operand edge cases, Enc-and-compare with plain and product expectations, keys at the edges of the form, and
    the same items through the pair ladder it replaces (set_r2l(0))."""
    import math
    rnd = random.Random(99)
    n_bits, kw = 2048, 64
    import os
    os.environ["ZKP_R2L_LANES"] = str(lanes)        # (read when the ctx is created: the kernel's lane geometry, 12 x 6 limbs by default)
    try:
        c = zkp.Context(0)
    finally:
        os.environ.pop("ZKP_R2L_LANES", None)
    try:
        c.set_geometry(9)
        for trial, n in enumerate((odd_modulus(rnd, 2048), odd_modulus(rnd, 2047), H.fixture_key()[2], odd_modulus(rnd, 1200))):
            nn = n * n
            count = 37 if trial else (1030 if lanes == 36 else 300)      # one launch of more than a wavefront per SIMD-quarter (five wavefronts per Enc: more items than workgroups), then small ones
            ms = [rnd.randrange(n) for _ in range(count)]
            rs = [rnd.getrandbits(n_bits) for _ in range(count)]
            ms[0], rs[0] = 0, 1
            ms[1], rs[1] = n - 1, n - 1
            ms[2], rs[2] = (1 << n_bits) - 1, (1 << n_bits) - 1
            ms[3], rs[3] = 5, 0
            ms[4], rs[4] = 7, n
            ms[5], rs[5] = n, n + 1
            nw = words(n, kw)
            mw = np.stack([words(v, kw) for v in ms]); rw = np.stack([words(v, kw) for v in rs])
            outs = []
            for mode in (2, 0):                                     # the ladder whenever it can run; never (the pair ladder / window ladder on n^2)
                c.set_r2l(mode)
                out = np.zeros((count, 2 * kw), np.uint32)
                c.paillier_enc(n_bits, count, nw, 0, mw, rw, out)
                assert c.last_geometry() == 9 and c.r2l_last() == (mode == 2), (trial, mode)
                assert c.r2l_lanes_last() == (lanes if mode == 2 else 0), (trial, mode, c.r2l_lanes_last())
                outs.append(out)
            assert np.array_equal(outs[0], outs[1])
            for i in range(count):
                if i >= 48 and i % 23:                              # (the two kernels agree on every item, above; Python checks the edge cases and a sample)
                    continue
                got = sum(int(w) << (32 * j) for j, w in enumerate(outs[0][i]))
                assert got == (1 + ms[i] * n) * pow(rs[i], n, nn) % nn, (trial, i)
            c.set_r2l(2)
            exp = outs[0].copy(); exp[6, 5] ^= 1
            ok = np.full(count, 9, np.uint8)
            c.paillier_enc_check(n_bits, count, nw, 0, mw, rw, None, None, exp, ok)
            assert c.r2l_last() and list(ok) == [0 if i == 6 else 1 for i in range(count)]
            bs = []
            while len(bs) < count:
                v = rnd.randrange(2, nn)
                if math.gcd(v, n) == 1:
                    bs.append(v)
            a_ = [(sum(int(w) << (32 * j) for j, w in enumerate(outs[0][i])) * pow(bs[i], -1, nn)) % nn for i in range(count)]
            a_[8] = (a_[8] + 1) % nn
            ok = np.full(count, 9, np.uint8)
            c.paillier_enc_check(n_bits, count, nw, 0, mw, rw, np.stack([words(v, 2 * kw) for v in a_]), np.stack([words(v, 2 * kw) for v in bs]), None, ok)
            assert c.r2l_last() and list(ok) == [0 if i == 8 else 1 for i in range(count)]
        # a key the form does not take (even): the launch behind the ladder does the work, the answers are those of the n^2-sized kernels
        c.set_r2l(1)
    finally:
        c.close()
