"""GPU parity: NiCorrectKeyProof::verify (correct_key_ni.rs:73-100) against the oracle."""
import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_bits,salt", [(1024, pm.SALT_STRING), (2048, bytes([90, 101, 110, 32, 71, 111, 32, 88])), (1024, b"\x00\x00ab"), (1024, b"")])
def test_correct_key_verify(ctx, oracle, n_bits, salt):
    kw = n_bits // 32
    keys = [H.test_key(n_bits, tag=t) for t in range(3)] if n_bits == 1024 else [H.fixture_key(), H.test_key(2048, 1)]
    ns, sigmas = [], []
    for p, q, n in keys:
        nl, sg = oracle.correct_key_ni_prove(n_bits, L.int_to_limbs(p, kw // 2), L.int_to_limbs(q, kw // 2), salt)
        ns.append(nl); sigmas.append(sg)
    # tampered sigma, sigma + n (same residue: still accepted), sigma = 0 row, modulus with a small factor, short key
    ns.append(ns[0]); bad = sigmas[0].copy(); bad[7, 1] ^= 4; sigmas.append(bad)
    ns.append(ns[1]); plus = sigmas[1].copy()
    v = L.limbs_to_int(plus[2]) + keys[1][2]
    if v.bit_length() <= n_bits:
        plus[2] = L.int_to_limbs(v, kw)
    sigmas.append(plus)
    ns.append(ns[0]); z = sigmas[0].copy(); z[0] = 0; sigmas.append(z)
    n_small = 6361 * H.gen_prime(pm.Drbg(b"sf-%d" % n_bits), n_bits - 16)
    ns.append(L.int_to_limbs(n_small, kw)); sigmas.append(L.ints_to_limbs(pm.correct_key_rho(n_small, salt), kw))
    pk, qk, nk = H.test_key(n_bits - 64, tag=9)      # n shorter than the ABI width: key_length drives the MGF length
    ns.append(L.int_to_limbs(nk, kw))
    sigmas.append(L.ints_to_limbs(pm.correct_key_proof(pk, qk, salt), kw))
    n_arr = np.stack(ns); s_arr = np.stack(sigmas)
    vo = oracle.correct_key_ni_verify(n_bits, n_arr, s_arr, salt)
    vg = np.full(len(ns), 7, np.uint8)
    ctx.correct_key_ni_verify(n_bits, len(ns), n_arr, s_arr, salt, vg)
    assert np.array_equal(vo, vg), (list(vo), list(vg))
    assert list(vo[:len(keys)]) == [zkp.VERDICT_ACCEPT] * len(keys)
    assert vo[len(keys)] == zkp.VERDICT_REJECT and vo[-1] == zkp.VERDICT_ACCEPT and vo[-2] == zkp.VERDICT_REJECT
