"""The constants of the ONE key of a shared-key call are kept across calls (csrc/kernels_modexp.hpp: the tag of a constants buffer;
csrc/zkp_api.hip: setup_tag): the set-up kernels return at once when they are handed the modulus their record was computed from.  What
can go wrong with that is a STALE record — another key's constants used for this call — so this file walks every way the record of a
buffer changes hands (another key, an even key, a call under per-proof keys, another width, released staging, another engine) and checks
every call's bytes against the C/GMP oracle, on each engine; and that switching the cache off changes no byte."""
import numpy as np
import pytest

import helpers as H
from helpers import L

zkp = H.zkp
pytestmark = pytest.mark.gpu

FIELDS = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")


@pytest.fixture(scope="module")
def kctx():
    c = zkp.Context(0)
    yield c
    c.close()


def prove_and_verify(c, oracle, seed, n, n_bits, B, expect_hit=None, tamper=True):
    """one prove + one verify call of B proofs under the key n: transcripts and verdicts against the oracle"""
    cases = H.build_range_case(seed, [n], n_bits, B)
    pb_o, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
    pb.n[:] = pb_o.n; pb.range[:] = pb_o.range; pb.ciphertext[:] = pb_o.ciphertext
    status = np.full(B, 9, np.uint8)
    c.range_ni_prove(pb.struct(), wt.struct(), None, None, status, device=False)
    if expect_hit is not None:
        hit = c.key_cache_state(0)[1]
        assert hit == expect_hit, (seed, "n^2 constants", hit, expect_hit)
    assert not status.any()
    oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
    for f in FIELDS:
        assert np.array_equal(getattr(pb_o, f), getattr(pb, f)), (seed, f)
    if tamper and B > 1:
        pb.resp_r1[B - 1, 5, 0] ^= 1
    v = np.full(B, 9, np.uint8)
    c.range_ni_verify(pb.struct(), v, device=False)
    vo = np.full(B, 9, np.uint8)
    oracle.range_ni_verify(pb.struct(), vo)
    assert np.array_equal(v, vo), (seed, v, vo)
    assert v[0] == 1
    return pb


@pytest.mark.parametrize("engine", [9, 18, 36])
def test_the_record_of_a_buffer_changes_hands(kctx, oracle, engine):
    c = kctx
    c.set_geometry(engine)
    c.set_enc_form("basen" if engine != 9 else "auto")
    c.set_key_cache(True)
    oracle.set_threads(min(16, oracle.max_threads()))
    K = H.fixture_key()[2]
    K2 = H.test_key(2048, 1)[2]
    K3 = H.test_key(2048, 2)[2]
    try:
        prove_and_verify(c, oracle, b"kc-a", K, 2048, 2)
        prove_and_verify(c, oracle, b"kc-b", K, 2048, 2, expect_hit=True)          # the same key again: the set-up returns early
        valid, hit, epoch = c.key_cache_state(0)
        assert valid and hit
        prove_and_verify(c, oracle, b"kc-c", K2, 2048, 2, expect_hit=False)        # another key: computed, and the bytes are K2's
        prove_and_verify(c, oracle, b"kc-d", K2, 2048, 3, expect_hit=True)
        prove_and_verify(c, oracle, b"kc-e", K, 2048, 2, expect_hit=False)         # back: K's record is gone, computed again
        # a key that differs from K in ONE high word only
        prove_and_verify(c, oracle, b"kc-f", K ^ (1 << 2000), 2048, 2, expect_hit=False)
        prove_and_verify(c, oracle, b"kc-g", K, 2048, 2, expect_hit=False)
        # an even key: rejected, never cached — and rejected again the second time
        for rep in range(2):
            pb = zkp.RangeBatch(2048, 1, 128, shared_key=True)
            pb.n[0] = L.int_to_limbs(K - 1, 64)
            wt = zkp.make_range_witness(2048, 1, 128)
            st = np.full(1, 9, np.uint8)
            c.range_ni_prove(pb.struct(), wt.struct(), None, None, st, device=False)
            assert st[0] == zkp.VERDICT_MALFORMED, st
            assert not c.key_cache_state(0)[0]
        prove_and_verify(c, oracle, b"kc-h", K, 2048, 2, expect_hit=False)
        prove_and_verify(c, oracle, b"kc-i", K, 2048, 2, expect_hit=True)
        # a call under per-proof keys writes several records into the same buffers: K's tag must not survive it
        keys = [K3, K2, K]
        cases = H.build_range_case(b"kc-multi", keys, 2048, 3, shared=False)
        pb_o, wt = H.fill_batch(cases, 2048, False, oracle)
        pbm = zkp.RangeBatch(2048, 3, 128, shared_key=False)
        pbm.n[:] = pb_o.n; pbm.range[:] = pb_o.range; pbm.ciphertext[:] = pb_o.ciphertext
        c.range_ni_prove(pbm.struct(), wt.struct(), None, None, None, device=False)
        oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
        for f in FIELDS:
            assert np.array_equal(getattr(pb_o, f), getattr(pbm, f)), ("per-proof keys", f)
        prove_and_verify(c, oracle, b"kc-j", K, 2048, 2, expect_hit=False)         # record 0 now was K3's
        prove_and_verify(c, oracle, b"kc-k", K, 2048, 2, expect_hit=True)
        # another width in between
        K1024 = H.test_key(1024, 0)[2]
        prove_and_verify(c, oracle, b"kc-l", K1024, 1024, 2)
        prove_and_verify(c, oracle, b"kc-m", K, 2048, 2, expect_hit=False)
        # the ctx gives its staging back (device blocks freed): whatever it keeps must still be right
        c.release_staging()
        prove_and_verify(c, oracle, b"kc-n", K, 2048, 2)
        prove_and_verify(c, oracle, b"kc-o", K2, 2048, 1, tamper=False)
        # plain Enc calls under one key go through the same records
        for n in (K, K, K2, K):
            count, kw = 5, 64
            mw = np.zeros((count, kw), np.uint32); mw[:, 0] = 7 + np.arange(count)
            rw = np.zeros((count, kw), np.uint32); rw[:, 0] = 3 + np.arange(count); rw[:, 40] = 0x1234567
            out = np.zeros((count, 2 * kw), np.uint32)
            c.paillier_enc(2048, count, L.int_to_limbs(n, kw), 0, mw, rw, out)
            for i in range(count):
                r = (3 + i) | (0x1234567 << (32 * 40))
                assert L.limbs_to_int(out[i]) == (1 + (7 + i) * n) * pow(r, n, n * n) % (n * n), (hex(n)[:12], i)
    finally:
        c.set_geometry(0)
        c.set_enc_form("auto")


def test_switching_the_cache_off_changes_no_byte(kctx, oracle):
    c = kctx
    K = H.fixture_key()[2]
    c.set_geometry(0); c.set_enc_form("auto")
    try:
        c.set_key_cache(True)
        a1 = prove_and_verify(c, oracle, b"kc-off", K, 2048, 2)
        a2 = prove_and_verify(c, oracle, b"kc-off", K, 2048, 2, expect_hit=True)
        c.set_key_cache(False)
        b1 = prove_and_verify(c, oracle, b"kc-off", K, 2048, 2)
        assert c.key_cache_state(0) == (False, False, 0) or not c.key_cache_state(0)[1]
        b2 = prove_and_verify(c, oracle, b"kc-off", K, 2048, 2)
        for f in FIELDS:
            assert np.array_equal(getattr(a1, f), getattr(b1, f)) and np.array_equal(getattr(a2, f), getattr(b2, f)), f
    finally:
        c.set_key_cache(True)


@pytest.mark.parametrize("engine", [9, 36])
def test_modexp_under_one_modulus(kctx, engine):
    """zkp_modexp_batch with a shared modulus sets up ONE modulus as well (the same constants buffer as the n^2 records, square = 0):
    M1, M1, M2, M1, then M1 at another width and n^2 of a Paillier call in between — against python's pow"""
    pm = H.pm
    c = kctx
    c.set_geometry(engine)
    try:
        d = pm.Drbg(b"kc-modexp")
        count, nl = 6, 64
        M1 = d.bits(2048) | 1 | (1 << 2047)
        M2 = d.bits(2048) | 1 | (1 << 2047)

        def run(mod, bits=2048, expect_hit=None):
            w = bits // 32
            bases = [d.below(mod) for _ in range(count)]
            exps = [d.bits(256) for _ in range(count)]
            b = L.ints_to_limbs(bases, w); e = L.ints_to_limbs(exps, 8); m = L.ints_to_limbs([mod], w)
            out = np.zeros_like(b)
            c.modexp(bits, 256, count, b, e, 8, m, 0, out)
            if expect_hit is not None:
                assert c.key_cache_state(0)[1] == expect_hit
            for i in range(count):
                assert L.limbs_to_int(out[i]) == pow(bases[i], exps[i], mod), (hex(mod)[:12], i)

        run(M1)
        run(M1, expect_hit=True)
        run(M2, expect_hit=False)
        run(M1, expect_hit=False)
        # the same words as a 4096-bit modulus (upper half zero): other parameters, computed
        run(M1, bits=4096, expect_hit=False)
        run(M1, expect_hit=False)
        # a Paillier call under n = M1 puts n^2's record where M1's was
        kw = 64
        mw = np.zeros((2, kw), np.uint32); mw[:, 0] = 5
        rw = np.zeros((2, kw), np.uint32); rw[:, 0] = 9
        out = np.zeros((2, 2 * kw), np.uint32)
        c.paillier_enc(2048, 2, L.int_to_limbs(M1, kw), 0, mw, rw, out)
        assert L.limbs_to_int(out[0]) == (1 + 5 * M1) * pow(9, M1, M1 * M1) % (M1 * M1)
        run(M1, expect_hit=False)
        run(M1, expect_hit=True)
    finally:
        c.set_geometry(0)
