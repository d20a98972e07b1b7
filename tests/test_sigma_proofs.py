"""ZeroProof (zero_enc_proof.rs) and CiphertextProof (correct_ciphertext.rs): oracle vs python model on the CPU,
HIP engine vs oracle on the GPU (SURVEY §8(f) rank 1 — compositions of the hot-path kernels)."""
import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp


def make_cases(n_bits, keys, B, seed):
    d = pm.Drbg(seed)
    kw = n_bits // 32
    rows = []
    for b in range(B):
        n = keys[b % len(keys)]
        x, r, xp, rp = d.below(n), d.below(n), d.below(n), d.below(n)
        rows.append(dict(n=n, x=x, r=r, xp=xp, rp=rp, c0=pm.enc(n, 0, r), c1=pm.enc(n, 1, r), cx=pm.enc(n, x, r)))
    arr = lambda k, w: L.ints_to_limbs([q[k] for q in rows], w)
    return rows, dict(n=arr("n", kw), x=arr("x", kw), r=arr("r", kw), xp=arr("xp", kw), rp=arr("rp", kw),
                      c0=arr("c0", 2 * kw), c1=arr("c1", 2 * kw), cx=arr("cx", 2 * kw))


def test_oracle_matches_python_model(oracle):
    n_bits, kw = 1024, 32
    keys = [H.test_key(1024, tag=t)[2] for t in range(2)]
    rows, a = make_cases(n_bits, keys, 4, b"sigma-cpu")
    z, aa = oracle.zero_proof_prove(n_bits, a["n"], kw, a["c0"], a["r"], a["rp"])
    z1, z2, cp = oracle.ciphertext_proof_prove(n_bits, a["n"], kw, a["cx"], a["x"], a["r"], a["xp"], a["rp"])
    for b, q in enumerate(rows):
        assert (L.limbs_to_int(z[b]), L.limbs_to_int(aa[b])) == pm.zero_proof_prove(q["n"], q["c0"], q["r"], q["rp"])
        assert (L.limbs_to_int(z1[b]), L.limbs_to_int(z2[b]), L.limbs_to_int(cp[b])) == pm.ciphertext_proof_prove(q["n"], q["cx"], q["x"], q["r"], q["xp"], q["rp"])
        assert pm.zero_proof_verify(q["n"], q["c0"], L.limbs_to_int(z[b]), L.limbs_to_int(aa[b]))
        assert pm.ciphertext_proof_verify(q["n"], q["cx"], L.limbs_to_int(z1[b]), L.limbs_to_int(z2[b]), L.limbs_to_int(cp[b]))
    assert list(oracle.zero_proof_verify(n_bits, a["n"], kw, a["c0"], z, aa)) == [1] * 4            # test_zero_proof, zero_enc_proof.rs:112-131
    assert list(oracle.ciphertext_proof_verify(n_bits, a["n"], kw, a["cx"], z1, z2, cp)) == [1] * 4  # test_ciphertext_proof, correct_ciphertext.rs:113-134
    # test_one_proof (zero_enc_proof.rs:134-155): c encrypts 1 -> rejected
    z_, a_ = oracle.zero_proof_prove(n_bits, a["n"], kw, a["c1"], a["r"], a["rp"])
    assert list(oracle.zero_proof_verify(n_bits, a["n"], kw, a["c1"], z_, a_)) == [0] * 4
    # test_bad_ciphertext_proof (correct_ciphertext.rs:137-162): witness r + 1 -> rejected
    r_bad = L.ints_to_limbs([q["r"] + 1 for q in rows], kw)
    z1b, z2b, cpb = oracle.ciphertext_proof_prove(n_bits, a["n"], kw, a["cx"], a["x"], r_bad, a["xp"], a["rp"])
    assert list(oracle.ciphertext_proof_verify(n_bits, a["n"], kw, a["cx"], z1b, z2b, cpb)) == [0] * 4


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,shared", [(1024, False), (2048, True), (4096, True)])
def test_gpu_sigma_proofs_match_oracle(ctx, oracle, n_bits, shared):
    kw = n_bits // 32
    if n_bits == 2048:
        keys = [H.fixture_key()[2]]
    else:
        keys = [H.test_key(n_bits, tag=t)[2] for t in range(1 if shared else 3)]
    B = 5 if n_bits < 4096 else 3
    rows, a = make_cases(n_bits, keys, B, b"sigma-gpu-%d" % n_bits)
    n_arr = a["n"][:1] if shared else a["n"]
    stride = 0 if shared else kw
    oracle.set_threads(min(8, oracle.max_threads()))
    # ---- ZeroProof: honest statement c0, dishonest statement c1 (encrypts 1)
    for cc, expect in ((a["c0"], 1), (a["c1"], 0)):
        zo, ao = oracle.zero_proof_prove(n_bits, n_arr, stride, cc, a["r"], a["rp"])
        zg = np.zeros_like(zo); ag = np.zeros_like(ao)
        ctx.zero_proof_prove(n_bits, B, n_arr, stride, cc, a["r"], a["rp"], zg, ag)
        assert np.array_equal(zo, zg) and np.array_equal(ao, ag)
        # tamper the last proof's z
        zt = zg.copy(); zt[B - 1, 0] ^= 1
        vo = oracle.zero_proof_verify(n_bits, n_arr, stride, cc, zt, ag)
        vg = np.full(B, 9, np.uint8)
        ctx.zero_proof_verify(n_bits, B, n_arr, stride, cc, zt, ag, vg)
        assert np.array_equal(vo, vg) and list(vo) == [expect] * (B - 1) + [0]
    # ---- CiphertextProof
    z1o, z2o, cpo = oracle.ciphertext_proof_prove(n_bits, n_arr, stride, a["cx"], a["x"], a["r"], a["xp"], a["rp"])
    z1g = np.zeros_like(z1o); z2g = np.zeros_like(z2o); cpg = np.zeros_like(cpo)
    ctx.ciphertext_proof_prove(n_bits, B, n_arr, stride, a["cx"], a["x"], a["r"], a["xp"], a["rp"], z1g, z2g, cpg)
    assert np.array_equal(z1o, z1g) and np.array_equal(z2o, z2g) and np.array_equal(cpo, cpg)
    z1t = z1g.copy(); z1t[0, 3] ^= 2            # tamper z1 of proof 0
    cpt = cpg.copy(); cpt[1, 5] ^= 1            # tamper c' of proof 1 (changes the challenge)
    vo = oracle.ciphertext_proof_verify(n_bits, n_arr, stride, a["cx"], z1t, z2g, cpt)
    vg = np.full(B, 9, np.uint8)
    ctx.ciphertext_proof_verify(n_bits, B, n_arr, stride, a["cx"], z1t, z2g, cpt, vg)
    assert np.array_equal(vo, vg) and list(vo) == [0, 0] + [1] * (B - 2)
