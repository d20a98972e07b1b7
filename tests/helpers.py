"""Shared deterministic input builders for the test-suite (inputs come from the repo's
SHA-256 counter DRBG, oracle/py_model.Drbg, so Python / C oracle / GPU see identical data)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import py_model as pm  # noqa: E402  (tests may import the oracle)

zkp = importlib.import_module("zk-paillier_amd")
L = zkp.limbs


def is_probable_prime(n, rounds=24):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    rnd = pm.Drbg(b"mr" + n.to_bytes((n.bit_length() + 7) // 8, "big")[:16])
    for _ in range(rounds):
        a = rnd.range(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_prime(drbg, bits):
    while True:
        c = drbg.bits(bits) | (1 << (bits - 1)) | 1
        if is_probable_prime(c):
            return c


_KEYS = {}


def test_key(bits, tag=0):
    """deterministic (p, q, n) with n of exactly `bits` bits"""
    key = (bits, tag)
    if key not in _KEYS:
        d = pm.Drbg(b"key-%d-%d" % (bits, tag))
        while True:
            p, q = gen_prime(d, bits // 2), gen_prime(d, bits // 2)
            n = p * q
            if n.bit_length() == bits and p != q:
                break
        _KEYS[key] = (p, q, n)
    return _KEYS[key]


def fixture_key():
    return pm.FIXTURE_P, pm.FIXTURE_Q, pm.FIXTURE_N


def build_range_case(seed: bytes, n_list, n_bits, batch, range_bits=256, shared=True, honest=True, ef=128):
    """python-int inputs for `batch` RangeProofNi proofs.
    honest=True: x < range/3 (range_proof_ni.rs:166) ; False: x in [100*range, 10000*range) (:184-187)."""
    d = pm.Drbg(seed)
    cases = []
    for b in range(batch):
        n = n_list[0] if shared else n_list[b]
        rng_q = d.bits(range_bits) | (1 << (range_bits - 1))
        r = d.below(n)
        x = d.below(rng_q // 3) if honest else d.range(100 * rng_q, 10000 * rng_q)
        w1, w2, r1, r2 = pm.sample_range_inputs(d, n, rng_q, ef)
        cases.append(dict(n=n, range=rng_q, x=x, r=r, w1=w1, w2=w2, r1=r1, r2=r2))
    return cases


def fill_batch(cases, n_bits, shared, oracle, ciphertexts=None):
    """-> (RangeBatch host, RangeWitness host).  ciphertext = Enc(x, r) computed with the oracle."""
    B = len(cases)
    ef = len(cases[0]["w1"])
    kw = n_bits // 32
    pb = zkp.RangeBatch(n_bits, B, ef, shared_key=shared)
    wt = zkp.make_range_witness(n_bits, B, ef)
    for b, c in enumerate(cases):
        if not shared or b == 0:
            pb.n[0 if shared else b] = L.int_to_limbs(c["n"], kw)
        pb.range[b] = L.int_to_limbs(c["range"], kw)
        wt.x[b] = L.int_to_limbs(c["x"], kw)
        wt.r[b] = L.int_to_limbs(c["r"], kw)
        for f in ("w1", "w2", "r1", "r2"):
            getattr(wt, f)[b] = L.ints_to_limbs(c[f], kw)
    nn = pb.n if not shared else np.repeat(pb.n, B, axis=0)
    pb.ciphertext[:] = oracle.paillier_enc(n_bits, np.ascontiguousarray(nn), kw, wt.x, wt.r)
    return pb, wt


def responses_from_batch(pb, b):
    """SoA row -> py_model response tuples for proof b"""
    out = []
    for i in range(pb.ef):
        if pb.resp_kind[b, i] == zkp.RESP_OPEN:
            out.append(("open", L.limbs_to_int(pb.resp_w1[b, i]), L.limbs_to_int(pb.resp_r1[b, i]),
                        L.limbs_to_int(pb.resp_w2[b, i]), L.limbs_to_int(pb.resp_r2[b, i])))
        else:
            out.append(("mask", int(pb.resp_j[b, i]), L.limbs_to_int(pb.resp_w1[b, i]), L.limbs_to_int(pb.resp_r1[b, i])))
    return out
