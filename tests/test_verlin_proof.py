"""VerlinProof (verlin_proof.rs:35-165): oracle vs python model (CPU); HIP engine vs oracle (GPU)."""
import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp


def make(n_bits, keys, B, seed, bad_last=True):
    d = pm.Drbg(seed)
    kw = n_bits // 32
    rows = []
    for b in range(B):
        n = keys[b % len(keys)]
        nn = n * n
        c, cp = pm.enc(n, d.below(n), d.below(n)), pm.enc(n, d.below(n), d.below(n))       # the two public ciphertexts (verlin_proof.rs tests)
        x, xp, xpp, rx = d.below(n), d.below(n), d.below(n), d.below(n)
        phi_x = pm.gen_phi(n, c, cp, x, xp, xpp, rx)
        if bad_last and b == B - 1:
            phi_x = (phi_x * 2) % nn                                                         # statement no longer matches the witness
        rows.append(dict(n=n, c=c, cp=cp, phi_x=phi_x, x=x, xp=xp, xpp=xpp, rx=rx, a=d.below(n), ap=d.below(n), app=d.below(n), ra=d.below(n)))
    arr = lambda k, w: L.ints_to_limbs([q[k] for q in rows], w)
    a = {k: arr(k, 2 * kw if k in ("c", "cp", "phi_x") else kw) for k in rows[0]}
    return rows, a


def test_oracle_matches_python_model(oracle):
    n_bits, kw = 1024, 32
    keys = [H.test_key(1024, tag=t)[2] for t in range(2)]
    rows, a = make(n_bits, keys, 3, b"verlin-cpu")
    phi_a, z, zp, zpp, rz = oracle.verlin_proof_prove(n_bits, a["n"], kw, a["c"], a["cp"], a["phi_x"], (a["x"], a["xp"], a["xpp"], a["rx"]), (a["a"], a["ap"], a["app"], a["ra"]))
    for b, q in enumerate(rows):
        exp = pm.verlin_prove(q["n"], q["c"], q["cp"], q["phi_x"], q["x"], q["xp"], q["xpp"], q["rx"], q["a"], q["ap"], q["app"], q["ra"])
        got = tuple(L.limbs_to_int(v[b]) for v in (phi_a, z, zp, zpp, rz))
        assert got == exp
        assert pm.verlin_verify(q["n"], q["c"], q["cp"], q["phi_x"], *got) == (b != 2)
    assert list(oracle.verlin_proof_verify(n_bits, a["n"], kw, a["c"], a["cp"], a["phi_x"], phi_a, z, zp, zpp, rz)) == [1, 1, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,shared", [(1024, False), (2048, True)])
def test_gpu_verlin_matches_oracle(ctx, oracle, n_bits, shared):
    kw = n_bits // 32
    keys = [H.fixture_key()[2]] if n_bits == 2048 else [H.test_key(1024, tag=t)[2] for t in range(3)]
    B = 4
    rows, a = make(n_bits, keys, B, b"verlin-gpu-%d" % n_bits)
    n_arr = a["n"][:1] if shared else a["n"]
    stride = 0 if shared else kw
    oracle.set_threads(min(8, oracle.max_threads()))
    wit, non = (a["x"], a["xp"], a["xpp"], a["rx"]), (a["a"], a["ap"], a["app"], a["ra"])
    o = oracle.verlin_proof_prove(n_bits, n_arr, stride, a["c"], a["cp"], a["phi_x"], wit, non)
    g = tuple(np.zeros_like(v) for v in o)
    ctx.verlin_proof_prove(n_bits, B, n_arr, stride, a["c"], a["cp"], a["phi_x"], wit, non, g)
    for vo, vg in zip(o, g):
        assert np.array_equal(vo, vg)
    zt = g[1].copy(); zt[0, 0] ^= 1                       # tamper z of proof 0
    vo = oracle.verlin_proof_verify(n_bits, n_arr, stride, a["c"], a["cp"], a["phi_x"], g[0], zt, g[2], g[3], g[4])
    vg = np.full(B, 9, np.uint8)
    ctx.verlin_proof_verify(n_bits, B, n_arr, stride, a["c"], a["cp"], a["phi_x"], g[0], zt, g[2], g[3], g[4], vg)
    assert np.array_equal(vo, vg) and list(vo) == [0, 1, 1, 0]
