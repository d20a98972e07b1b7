"""The constants of the hot path as READ from the reference tree (tools/pin_reference_constants.py -> tests/golden/reference_constants.json)
against every place this repo restates them: the C/GMP oracle (which COMPUTES the primorial), the Python model, the package's exports, the
boundary header, the C++ host layer, the synthetic-input generator, and — under -m gpu — the device's small-prime table.

This does not pin the oracle's arithmetic to the reference (the reference cannot be compiled here: no rustc) — it
removes the last values that were recalled rather than read."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from helpers import pm, L

zkp = H.zkp
GOLD = os.path.join(H.ROOT, "tests", "golden", "reference_constants.json")
K = json.load(open(GOLD))["constants"]


def val(name):
    return K[name]["value"]


def test_committed_file_matches_the_reference_tree_when_it_is_here():
    if not os.path.isdir("/root/reference/src/zkproofs"):
        pytest.skip("the reference tree exists in the build container only")
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "pin_reference_constants.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_primorial_the_oracle_computes_is_the_string_the_reference_parses(oracle):
    """correct_key_ni.rs:26,87 — oracle/zkp_oracle.c: primorial_init multiplies the primes below 6370"""
    P = int(val("correct_key_ni.P"))
    assert oracle.primorial_decimal() == val("correct_key_ni.P")
    assert pm.primorial() == P
    # and P is exactly the square-free product of ALL primes below 6370: what the device's divisibility test assumes
    primes = pm.primes_below(pm.ALPHA)
    rest = P
    for p in primes:
        assert rest % p == 0
        rest //= p
        assert rest % p != 0
    assert rest == 1 and len(primes) == 830 and P.bit_length() == K["correct_key_ni.P"]["bits"] == 9095


def test_small_constants_everywhere():
    assert bytes(val("correct_key_ni.SALT_STRING")) == pm.SALT_STRING == b"KZen"
    assert val("correct_key_ni.M2") == pm.M2 == zkp.CORRECT_KEY_M2 == 11
    assert val("correct_key_ni.DIGEST_SIZE") == pm.DIGEST_SIZE == 256
    assert val("range_proof_ni.SECURITY_PARAMETER") == pm.SECURITY_PARAMETER == zkp.SECURITY_PARAMETER == 128
    assert val("correct_message.B") == pm.CM_B
    header = open(os.path.join(H.ROOT, "include", "zkp_hip.h")).read()
    assert int(re.search(r"#define ZKP_SECURITY_PARAMETER (\d+)", header).group(1)) == val("range_proof_ni.SECURITY_PARAMETER")
    assert int(re.search(r"#define ZKP_CORRECT_KEY_M2 (\d+)", header).group(1)) == val("correct_key_ni.M2")
    host = open(os.path.join(H.ROOT, "zk-paillier_amd", "host", "zkproofs.hpp")).read()
    assert [int(v) for v in re.findall(r"STATISTICAL_ERROR_FACTOR = (\d+);", host)] == [val("range_proof.STATISTICAL_ERROR_FACTOR"), val("correct_key.STATISTICAL_ERROR_FACTOR")]
    assert int(re.search(r"DIGEST_SIZE = (\d+);", host).group(1)) == val("correct_key_ni.DIGEST_SIZE")
    # wi_dlog_proof.rs:53-54: the nonce is sample_below(2^(K + K_PRIME + SAMPLE_S)); every layer here spells that 512
    nonce_bits = val("wi_dlog_proof.K") + val("wi_dlog_proof.K_PRIME") + val("wi_dlog_proof.SAMPLE_S")
    assert nonce_bits == 512
    assert "BigInt::pow2(512)" in host
    assert "modexp_core<G>(c, 512," in open(os.path.join(H.ROOT, "zk-paillier_amd", "csrc", "zkp_api_proofs.inc")).read()
    oc = open(os.path.join(H.ROOT, "oracle", "zkp_oracle.c")).read()
    assert "v < 6370" in oc and pm.ALPHA == 6370


def test_fixture_keypair():
    """range_proof_ni.rs:141-145 (the same pair in benches/all.rs:73-77): the key of BASELINE configs[0..2]"""
    p, q = int(val("range_proof_ni.tests.test_keypair.p")), int(val("range_proof_ni.tests.test_keypair.q"))
    assert (p, q) == (pm.FIXTURE_P, pm.FIXTURE_Q) == H.fixture_key()[:2]
    assert (int(val("benches.test_keypair.p")), int(val("benches.test_keypair.q"))) == (p, q)
    assert p.bit_length() == q.bit_length() == 1024 and (p * q).bit_length() == 2048
    import importlib
    synth = importlib.import_module("zk-paillier_amd.synth")
    assert synth.BENCH_N == p * q
    import math
    assert math.gcd(int(val("correct_key_ni.P")), p * q) == 1


@pytest.mark.gpu
def test_device_small_prime_table_is_the_factor_set_of_P(oracle):
    """NiCorrectKeyProof::verify's gcd(P, n) == 1 runs on the device as 830 trial divisions (csrc/kernels_proofs.hpp): every prime factor
    of the reference's P, and no other number, must make a key fail.  Keys n = f * Q for every prime f below 6400: the verdict of a
    CORRECT proof for that key (sigma from the factorisation) is accept exactly when f does not divide P — and equal to the oracle's."""
    P = int(val("correct_key_ni.P"))
    n_bits, kw = 1024, 32
    Q = H.gen_prime(pm.Drbg(b"ck-table-Q"), 1000)
    fs = [f for f in range(3, 6400, 2) if H.is_probable_prime(f)]
    import math
    fs = [f for f in fs if math.gcd(f * Q, (f - 1) * (Q - 1)) == 1]          # keys for which n-th roots exist (gcd(n, phi) = 1)
    assert sum(1 for f in fs if P % f == 0) > 780 and sum(1 for f in fs if P % f) >= 3
    ns = [f * Q for f in fs]
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(f, Q, pm.SALT_STRING), kw) for f in fs])
    n_arr = L.ints_to_limbs(ns, kw)
    ctx = zkp.Context(0)
    try:
        for geometry in (36, 9):
            ctx.set_geometry(geometry)
            v = np.full(len(fs), 9, np.uint8)
            ctx.correct_key_ni_verify(n_bits, len(fs), n_arr, sig, pm.SALT_STRING, v)
            want = [zkp.VERDICT_REJECT if P % f == 0 else zkp.VERDICT_ACCEPT for f in fs]
            assert list(v) == want, [f for f, a, b in zip(fs, v, want) if a != b][:10]
        vo = oracle.correct_key_ni_verify(n_bits, n_arr, sig, pm.SALT_STRING)
        assert list(vo) == want
    finally:
        ctx.close()
