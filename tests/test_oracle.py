"""CPU tests: the C/GMP oracle against the independent pure-Python model (oracle/py_model.py)."""
import hashlib

import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp


def test_sha256_matches_hashlib(oracle):
    d = pm.Drbg(b"sha")
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 131328):
        msg = d.bytes(n)
        assert oracle.sha256(msg) == hashlib.sha256(msg).digest()


def test_modexp_and_modmul_small(oracle):
    d = pm.Drbg(b"modexp")
    kw = 64
    mods, bases, exps = [], [], []
    for i in range(6):
        m = d.bits(2048) | 1 | (1 << 2047)
        mods.append(m); bases.append(d.below(m)); exps.append(d.bits(2048))
    out = oracle.modexp(2048, 2048, L.ints_to_limbs(bases, kw), L.ints_to_limbs(exps, kw), kw, L.ints_to_limbs(mods, kw), kw)
    assert L.limbs_to_ints(out) == [pow(b, e, m) for b, e, m in zip(bases, exps, mods)]
    out = oracle.modmul(2048, L.ints_to_limbs(bases, kw), L.ints_to_limbs(exps, kw), L.ints_to_limbs(mods, kw), kw)
    assert L.limbs_to_ints(out) == [(b * e) % m for b, e, m in zip(bases, exps, mods)]


def test_enc_fixture_key(oracle):
    _, _, n = H.fixture_key()
    d = pm.Drbg(b"enc")
    ms = [0, 1, d.bits(256), n - 1]
    rs = [d.below(n) for _ in ms]
    out = oracle.paillier_enc(2048, L.ints_to_limbs([n], 64), 0, L.ints_to_limbs(ms, 64), L.ints_to_limbs(rs, 64))
    assert L.limbs_to_ints(out) == [pm.enc(n, m, r) for m, r in zip(ms, rs)]


@pytest.mark.parametrize("honest", [True, False])
def test_range_ni_against_python_model(oracle, honest):
    """full prove + verify on a 512-bit key (cheap in pure Python), 1024-bit ABI width"""
    n_bits = 1024
    _, _, n = H.test_key(512)
    cases = H.build_range_case(b"rp-%d" % honest, [n], n_bits, 2, honest=honest)
    pb, wt = H.fill_batch(cases, n_bits, True, oracle)
    e = np.zeros((2, 32), np.uint8); elen = np.zeros(2, np.uint8); st = np.zeros(2, np.uint8)
    oracle.range_ni_prove(pb.struct(), wt.struct(), e, elen, st)
    verdict = np.zeros(2, np.uint8)
    oracle.range_ni_verify(pb.struct(), verdict)
    for b, c in enumerate(cases):
        ct = pm.enc(n, c["x"], c["r"])
        assert L.limbs_to_int(pb.ciphertext[b]) == ct
        proof = pm.range_ni_prove(n, c["range"], ct, c["x"], c["r"], c["w1"], c["w2"], c["r1"], c["r2"])
        assert L.limbs_to_ints(pb.c1[b]) == proof["c1"]
        assert L.limbs_to_ints(pb.c2[b]) == proof["c2"]
        assert bytes(e[b, :elen[b]]) == proof["e"]
        assert H.responses_from_batch(pb, b) == proof["responses"]
        assert bool(verdict[b] == zkp.VERDICT_ACCEPT) == pm.range_ni_verify(proof, n, ct)
        assert (verdict[b] == zkp.VERDICT_ACCEPT) == honest   # range_proof_ni.rs:163-199


def test_range_verify_rejections(oracle):
    """tampering cases: each must flip the verdict exactly as the Python model says"""
    n_bits = 1024
    _, _, n = H.test_key(512)
    cases = H.build_range_case(b"tamper", [n], n_bits, 1)
    pb, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None)
    base = {f: getattr(pb, f).copy() for f in ("resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2", "c1", "c2")}
    ct = L.limbs_to_int(pb.ciphertext[0])
    mask_rows = [i for i in range(128) if pb.resp_kind[0, i] == zkp.RESP_MASK]
    open_rows = [i for i in range(128) if pb.resp_kind[0, i] == zkp.RESP_OPEN]

    def check():
        v = np.zeros(1, np.uint8)
        oracle.range_ni_verify(pb.struct(), v)
        proof = dict(n=n, range=cases[0]["range"], ciphertext=ct, c1=L.limbs_to_ints(pb.c1[0]), c2=L.limbs_to_ints(pb.c2[0]),
                     responses=H.responses_from_batch(pb, 0), error_factor=128)
        assert bool(v[0] == zkp.VERDICT_ACCEPT) == pm.range_ni_verify(proof, n, ct)
        return v[0]

    def restore():
        for f, a in base.items():
            getattr(pb, f)[:] = a

    assert check() == zkp.VERDICT_ACCEPT
    pb.resp_r1[0, mask_rows[0], 0] ^= 1; assert check() == zkp.VERDICT_REJECT; restore()
    pb.resp_w2[0, open_rows[0], 0] ^= 1; assert check() == zkp.VERDICT_REJECT; restore()
    pb.resp_kind[0, open_rows[1]] = zkp.RESP_MASK; assert check() == zkp.VERDICT_REJECT; restore()
    # j other than 1 selects c2 (range_proof.rs:324-328): flipping j of a j=2 row to 7 keeps the proof valid
    j2 = [i for i in mask_rows if pb.resp_j[0, i] == 2]
    if j2:
        pb.resp_j[0, j2[0]] = 7; assert check() == zkp.VERDICT_ACCEPT; restore()
    assert check() == zkp.VERDICT_ACCEPT


def test_correct_key_ni(oracle):
    p, q, n = H.test_key(1024)
    kw = 32
    for salt in (pm.SALT_STRING, bytes([90, 101, 110, 32, 71, 111, 32, 88]), b"\x00\x00ab"):
        nl, sigma = oracle.correct_key_ni_prove(1024, L.int_to_limbs(p, kw // 2), L.int_to_limbs(q, kw // 2), salt)
        assert L.limbs_to_int(nl) == n
        assert L.limbs_to_ints(sigma) == pm.correct_key_proof(p, q, salt)
        assert L.limbs_to_ints(oracle.correct_key_rho(1024, nl, salt)) == pm.correct_key_rho(n, salt)
        v = oracle.correct_key_ni_verify(1024, nl[None, :], sigma[None, :, :], salt)
        assert v[0] == zkp.VERDICT_ACCEPT and pm.correct_key_verify(L.limbs_to_ints(sigma), n, salt)
        bad = sigma.copy(); bad[3, 0] ^= 2
        assert oracle.correct_key_ni_verify(1024, nl[None, :], bad[None, :, :], salt)[0] == zkp.VERDICT_REJECT
    # n with a factor below 6370 fails the gcd test (correct_key_ni.rs:87-88,95)
    n_bad = 6361 * H.gen_prime(pm.Drbg(b"smallfactor"), 1000)
    sig = L.ints_to_limbs([pow(r, 1, n_bad) for r in pm.correct_key_rho(n_bad, pm.SALT_STRING)], kw)
    assert oracle.correct_key_ni_verify(1024, L.int_to_limbs(n_bad, kw)[None, :], sig[None], pm.SALT_STRING)[0] == zkp.VERDICT_REJECT


def test_primorial_constant():
    """P of correct_key_ni.rs:26 is the product of the 830 primes below 6370 (9095 bits)"""
    assert len(pm.primes_below(6370)) == 830 and pm.primorial().bit_length() == 9095


def test_dlog(oracle):
    p, q, N = H.test_key(1024, tag=1)
    kw, yw = 32, 24
    d = pm.Drbg(b"dlog")
    g = d.range(2, N - 1)
    s = d.bits(256)
    ni_good = pow(pow(g, -1, N), s, N)     # wi_dlog_proof.rs:130-131
    ni_bad = pow(g, s, N)                  # :159
    r = d.bits(512)
    for ni, ok in ((ni_good, True), (ni_bad, False)):
        x, y = oracle.dlog_prove(1024, 768, *(L.ints_to_limbs([v], kw) for v in (N, g, ni)), L.ints_to_limbs([s], 8), L.ints_to_limbs([r], 16))
        px, py = pm.dlog_prove(N, g, ni, s, r)
        assert L.limbs_to_int(x[0]) == px and L.limbs_to_int(y[0]) == py
        v = oracle.dlog_verify(1024, 768, *(L.ints_to_limbs([v], kw) for v in (N, g, ni)), x, y)
        assert bool(v[0] == zkp.VERDICT_ACCEPT) == ok == pm.dlog_verify(px, py, N, g, ni)
    # gcd(g, N) != 1 is a panic in the reference (:72) -> malformed
    v = oracle.dlog_verify(1024, 768, *(L.ints_to_limbs([v], kw) for v in (N, p, ni_good)), x, y)
    assert v[0] == zkp.VERDICT_MALFORMED


def test_challenge_leading_zero_byte(oracle):
    """N2: a digest with a leading 00 byte yields a 31-byte challenge whose bit 0 is the MSB of byte 1"""
    n_bits, kw = 1024, 32
    _, _, n = H.test_key(512)
    nl = L.int_to_limbs(n, kw)
    d = pm.Drbg(b"lz")
    ef = 2
    for ctr in range(4000):
        c1 = [d.bits(64) for _ in range(ef)]; c2 = [d.bits(64) for _ in range(ef)]
        if hashlib.sha256(b"".join(pm.to_bytes(v) for v in [n] + c1 + c2)).digest()[0] == 0:
            break
    else:
        pytest.skip("no leading-zero digest found")
    e = oracle.fs_challenge(n_bits, ef, nl, L.ints_to_limbs(c1, 2 * kw), L.ints_to_limbs(c2, 2 * kw))
    assert e == pm.fs_challenge(n, c1, c2) and len(e) < 32
