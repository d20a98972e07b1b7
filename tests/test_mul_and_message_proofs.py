"""mod_inv, MulProof (multiplication_proof.rs) and CorrectMessageProof (correct_message.rs) — SURVEY §8(f) rank 4.
CPU: the C oracle against the pure-Python model (and the reference's own accept / reject behaviours,
multiplication_proof.rs:172-290, correct_message.rs:169-200).  GPU: the HIP engine against the oracle, byte-exact."""
import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

MALFORMED = zkp.VERDICT_MALFORMED


# ------------------------------------------------------------------ mod_inv
def inv_cases(mod, kw, seed, count):
    d = pm.Drbg(seed)
    vals = [0, 1, 2, mod - 1, mod - 2, (mod + 1) // 2, 1 << 31, 1 << 32, (1 << 64) - 1, 1 << (mod.bit_length() - 1)]
    # a == M modulo 2^32 / 2^64 / 2^96: the first difference has whole zero low words (the unfused path of k_modinv)
    vals += [mod - (3 << 32), mod - (5 << 64), mod - (7 << 96), mod - (d.below(1 << 200) << 64)]
    vals += [d.below(mod) for _ in range(count - len(vals))]
    return vals


def check_inverse(vals, mod, out, st):
    for i, a in enumerate(vals):
        inv = pm.mod_inv(a, mod) if a < mod and mod % 2 == 1 and mod >= 3 else None
        if a >= mod or mod % 2 == 0 or mod < 3:
            assert st[i] == zkp.INV_DOMAIN, i
        elif inv is None:
            assert st[i] == zkp.INV_NONE, i
        else:
            assert st[i] == zkp.INV_OK and L.limbs_to_int(out[i]) == inv, i
        if st[i] != zkp.INV_OK:
            assert not out[i].any()


def test_oracle_modinv(oracle):
    p_, q_, n = H.test_key(512)
    nn = n * n
    kw = 64
    vals = inv_cases(nn, kw, b"inv-cpu", 40) + [p_, q_ * 7, n, n * p_, nn + 5 if (nn + 5).bit_length() <= 2048 else 3]
    a = L.ints_to_limbs(vals, kw)
    out, st = oracle.modinv(2048, a, L.int_to_limbs(nn, kw)[None, :], 0)
    check_inverse(vals, nn, out, st)
    assert st[vals.index(p_)] == zkp.INV_NONE and st[0] == zkp.INV_NONE
    # per-item moduli, one of them even
    mods = [nn, nn - 1, 3 * 5 * 7, 3]
    vals2 = [5, 5, 10, 2]
    out, st = oracle.modinv(2048, L.ints_to_limbs(vals2, kw), L.ints_to_limbs(mods, kw), kw)
    assert list(st) == [zkp.INV_OK, zkp.INV_DOMAIN, zkp.INV_NONE, zkp.INV_OK]
    assert L.limbs_to_int(out[3]) == 2


# ------------------------------------------------------------------ MulProof
MUL_IN = ("e_a", "e_b", "e_c", "a", "b", "r_a", "r_b", "r_c", "d", "r_d")


def mul_cases(n_bits, keys, B, seed, honest=True):
    d = pm.Drbg(seed)
    kw = n_bits // 32
    rows = []
    for i in range(B):
        n = keys[i % len(keys)]
        a, b = d.below(n), d.below(n)
        c = a * b % n if honest else (a * b + 1) % n
        r_a, r_b, r_c, dd, r_d = (d.below(n) for _ in range(5))
        rows.append(dict(n=n, a=a, b=b, c=c, r_a=r_a, r_b=r_b, r_c=r_c, d=dd, r_d=r_d,
                         e_a=pm.enc(n, a, r_a), e_b=pm.enc(n, b, r_b), e_c=pm.enc(n, c, r_c)))
    arr = {k: L.ints_to_limbs([q[k] for q in rows], 2 * kw if k.startswith("e_") else kw) for k in ("n",) + MUL_IN}
    return rows, arr


def test_oracle_mul_proof_matches_python_model(oracle):
    n_bits, kw = 1024, 32
    keys = [H.test_key(1024, tag=t)[2] for t in range(2)]
    rows, a = mul_cases(n_bits, keys, 4, b"mul-cpu")
    f, z1, z2, e_d, e_db, st = oracle.mul_proof_prove(n_bits, a["n"], kw, *[a[k] for k in MUL_IN])
    assert list(st) == [0] * 4
    for i, q in enumerate(rows):
        exp = pm.mul_proof_prove(q["n"], q["e_a"], q["e_b"], q["e_c"], q["a"], q["b"], q["r_a"], q["r_b"], q["r_c"], q["d"], q["r_d"])
        got = tuple(L.limbs_to_int(x[i]) for x in (f, z1, z2, e_d, e_db))
        assert got == exp
        assert pm.mul_proof_verify(q["n"], q["e_a"], q["e_b"], q["e_c"], *got)
    assert list(oracle.mul_proof_verify(n_bits, a["n"], kw, a["e_a"], a["e_b"], a["e_c"], f, z1, z2, e_d, e_db)) == [1] * 4   # test_mul_proof :172-229
    # test_bad_mul_proof (:232-290): c = a*b + 1
    rows, a = mul_cases(n_bits, keys, 3, b"mul-bad", honest=False)
    f, z1, z2, e_d, e_db, st = oracle.mul_proof_prove(n_bits, a["n"], kw, *[a[k] for k in MUL_IN])
    assert list(oracle.mul_proof_verify(n_bits, a["n"], kw, a["e_a"], a["e_b"], a["e_c"], f, z1, z2, e_d, e_db)) == [0] * 3
    # r_c sharing a factor with n: mod_inv(...).unwrap() panics in prove (:95)
    p_, q_, n = H.test_key(1024, tag=0)
    rows, a = mul_cases(n_bits, [n], 2, b"mul-panic")
    a["r_c"][1] = L.int_to_limbs(p_, kw)
    f, z1, z2, e_d, e_db, st = oracle.mul_proof_prove(n_bits, a["n"], kw, *[a[k] for k in MUL_IN])
    assert list(st) == [0, MALFORMED]
    with pytest.raises(pm.Panic):
        q = rows[1]
        pm.mul_proof_prove(q["n"], q["e_a"], q["e_b"], q["e_c"], q["a"], q["b"], q["r_a"], q["r_b"], p_, q["d"], q["r_d"])
    # e_db = a multiple of p: the verifier's mod_inv(...).unwrap() panics (:135)
    e_db2 = e_db.copy(); e_db2[0] = L.int_to_limbs(p_ * 12345, 2 * kw)
    assert list(oracle.mul_proof_verify(n_bits, a["n"], kw, a["e_a"], a["e_b"], a["e_c"], f, z1, z2, e_d, e_db2))[0] == MALFORMED


# ------------------------------------------------------------------ CorrectMessageProof
def cm_cases(n_bits, keys, B, K, seed, pick=None):
    d = pm.Drbg(seed)
    kw = n_bits // 32
    rows = []
    for i in range(B):
        n = keys[i % len(keys)]
        valid = [d.below(1 << 64) + 3 for _ in range(K)]
        idx = (i % K) if pick is None else pick
        msg = valid[idx] if idx is not None and idx >= 0 else valid[0] + 1
        rows.append(dict(n=n, valid=valid, msg=msg, r=d.below(n), w=d.below(n), e_sim=[d.bits(256) for _ in range(K - 1)],
                         z_sim=[d.below(n) for _ in range(K - 1)]))
    arr = dict(n=L.ints_to_limbs([q["n"] for q in rows], kw),
               valid=np.stack([L.ints_to_limbs(q["valid"], kw) for q in rows]),
               msg=L.ints_to_limbs([q["msg"] for q in rows], kw), r=L.ints_to_limbs([q["r"] for q in rows], kw),
               w=L.ints_to_limbs([q["w"] for q in rows], kw),
               e_sim=np.stack([L.ints_to_limbs(q["e_sim"], 8) if K > 1 else np.zeros((0, 8), np.uint32) for q in rows]),
               z_sim=np.stack([L.ints_to_limbs(q["z_sim"], kw) if K > 1 else np.zeros((0, kw), np.uint32) for q in rows]))
    return rows, arr


@pytest.mark.parametrize("K", [1, 3])
def test_oracle_correct_message_matches_python_model(oracle, K):
    n_bits, kw = 1024, 32
    keys = [H.test_key(1024, tag=t)[2] for t in range(2)]
    B = 4
    rows, a = cm_cases(n_bits, keys, B, K, b"cm-cpu-%d" % K)
    ct, e_vec, z_vec, a_vec, st = oracle.correct_message_prove(n_bits, K, a["n"], kw, a["valid"], a["msg"], a["r"], a["e_sim"], a["z_sim"], a["w"])
    assert list(st) == [0] * B
    for b, q in enumerate(rows):
        exp = pm.correct_message_prove(q["n"], q["valid"], q["msg"], q["r"], q["e_sim"], q["z_sim"], q["w"])
        assert L.limbs_to_int(ct[b]) == exp[0]
        assert [L.limbs_to_int(x) for x in e_vec[b]] == exp[1]
        assert [L.limbs_to_int(x) for x in z_vec[b]] == exp[2]
        assert [L.limbs_to_int(x) for x in a_vec[b]] == exp[3]
        assert pm.correct_message_verify(q["n"], q["valid"], *exp)
    v = oracle.correct_message_verify(n_bits, K, a["n"], kw, a["valid"], ct, e_vec, z_vec, a_vec)
    assert list(v) == [1] * B                                        # test_correct_message_zk_proof :169-181
    # a tampered z: rejected; a tampered e: the assert_eq! panics
    z2 = z_vec.copy(); z2[0, 0, 0] ^= 1
    e2 = e_vec.copy(); e2[1, 0, 0] ^= 1
    assert oracle.correct_message_verify(n_bits, K, a["n"], kw, a["valid"], ct, e_vec, z2, a_vec)[0] == 0
    assert oracle.correct_message_verify(n_bits, K, a["n"], kw, a["valid"], ct, e2, z_vec, a_vec)[1] == MALFORMED


def test_oracle_correct_message_wrong_message(oracle):
    """test_incorrect_message_zk_proof (correct_message.rs:184-200, #[should_panic]): the encrypted message is not in the list"""
    n_bits, kw, K = 1024, 32, 3
    keys = [H.test_key(1024)[2]]
    rows, a = cm_cases(n_bits, keys, 2, K, b"cm-bad", pick=-1)
    ct, e_vec, z_vec, a_vec, st = oracle.correct_message_prove(n_bits, K, a["n"][:1], 0, a["valid"], a["msg"], a["r"], a["e_sim"], a["z_sim"], a["w"])
    assert list(st) == [MALFORMED] * 2
    with pytest.raises(pm.Panic):
        q = rows[0]
        pm.correct_message_prove(q["n"], q["valid"], q["msg"], q["r"], q["e_sim"], q["z_sim"], q["w"])


# ================================================================== GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("mod_bits", [2048, 4096, 8192])
def test_gpu_modinv_matches_oracle(ctx, oracle, mod_bits):
    kw = mod_bits // 32
    p_, q_, n = H.test_key(mod_bits // 2)
    nn = n * n
    vals = inv_cases(nn, kw, b"inv-gpu-%d" % mod_bits, 150) + [p_, q_ * 7, n, n * p_, p_ * p_, nn - n]
    # values that stress the word-level paths: long runs of trailing zeros (whole zero words), tiny values
    vals += [1 << 32, 1 << 64, 3 << 96, (d_ := pm.Drbg(b"z")).below(nn) >> 70 << 70, 5, nn - (1 << 40)]
    a = L.ints_to_limbs(vals, kw)
    m = L.int_to_limbs(nn, kw)[None, :]
    oo, so = oracle.modinv(mod_bits, a, m, 0)
    og = np.full_like(a, 0xA5A5A5A5); sg = np.full(len(vals), 9, np.uint8)
    ctx.modinv(mod_bits, len(vals), a, m, 0, og, sg)
    assert np.array_equal(so, sg) and np.array_equal(oo, og)
    check_inverse(vals, nn, og, sg)
    # per-item moduli, including an even one and a = modulus
    mods = [nn, nn - 1, 3 * 5 * 7, 3, n * 3, (1 << (mod_bits - 1)) + 1]
    vals2 = [5, 5, 10, 2, p_, 12345]
    a2 = L.ints_to_limbs(vals2, kw); m2 = L.ints_to_limbs(mods, kw)
    oo, so = oracle.modinv(mod_bits, a2, m2, kw)
    og = np.zeros_like(a2); sg = np.full(len(vals2), 9, np.uint8)
    ctx.modinv(mod_bits, len(vals2), a2, m2, kw, og, sg)
    assert np.array_equal(so, sg) and np.array_equal(oo, og)


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,shared", [(1024, False), (2048, True)])
def test_gpu_mul_proof_matches_oracle(ctx, oracle, n_bits, shared):
    kw = n_bits // 32
    if n_bits == 2048:
        pq = [H.fixture_key()]
    else:
        pq = [H.test_key(n_bits, tag=t) for t in range(1 if shared else 3)]
    keys = [k[2] for k in pq]
    B = 6
    oracle.set_threads(min(8, oracle.max_threads()))
    rows, a = mul_cases(n_bits, keys, B, b"mul-gpu-%d" % n_bits)
    bad_rows, bad = mul_cases(n_bits, keys, B, b"mul-gpu-bad-%d" % n_bits, honest=False)
    for k in MUL_IN:                      # proofs 4 and 5 are about a false statement (test_bad_mul_proof)
        a[k][4:] = bad[k][4:]
    a["r_c"][3] = L.int_to_limbs(pq[3 % len(pq)][0], kw)   # r_c = p: the prover's mod_inv has no result
    n_arr = a["n"][:1] if shared else a["n"]
    stride = 0 if shared else kw
    ins = [a[k] for k in MUL_IN]
    fo, z1o, z2o, edo, edbo, so = oracle.mul_proof_prove(n_bits, n_arr, stride, *ins)
    fg = np.full((B, kw), 7, np.uint32); z1g, z2g, edg, edbg = (np.full((B, 2 * kw), 7, np.uint32) for _ in range(4)); sg = np.full(B, 9, np.uint8)
    ctx.mul_proof_prove(n_bits, B, n_arr, stride, *ins, fg, z1g, z2g, edg, edbg, sg)
    assert list(so) == [0, 0, 0, MALFORMED, 0, 0] and np.array_equal(so, sg)
    for name, x, y in (("f", fo, fg), ("z1", z1o, z1g), ("z2", z2o, z2g), ("e_d", edo, edg), ("e_db", edbo, edbg)):
        assert np.array_equal(x, y), name
    # verify: honest, panicked-in-prove (zeros), false statement; then tampered copies
    def both(f, z1, z2, e_d, e_db):
        vo = oracle.mul_proof_verify(n_bits, n_arr, stride, a["e_a"], a["e_b"], a["e_c"], f, z1, z2, e_d, e_db)
        vg = np.full(B, 9, np.uint8)
        ctx.mul_proof_verify(n_bits, B, n_arr, stride, a["e_a"], a["e_b"], a["e_c"], f, z1, z2, e_d, e_db, vg)
        assert np.array_equal(vo, vg), (vo, vg)
        return list(vo)
    assert both(fo, z1o, z2o, edo, edbo) == [1, 1, 1, 0, 0, 0]
    f2 = fo.copy(); f2[0, 0] ^= 1
    z12 = z1o.copy(); z12[1, 5] ^= 4
    z22 = z2o.copy(); z22[2, kw] ^= 1
    assert both(f2, z12, z22, edo, edbo) == [0, 0, 0, 0, 0, 0]
    # e_db sharing a factor with n: the verifier's unwrap panics; e_d out of range (>= n^2): hashed as is, used modulo n^2
    edb2 = edbo.copy(); edb2[0] = L.int_to_limbs(pq[0][0] * 98765, 2 * kw)
    ed2 = edo.copy(); ed2[1] = L.int_to_limbs(L.limbs_to_int(edo[1]) + keys[1 % len(keys)] ** 2, 2 * kw) if n_bits == 1024 else edo[1]
    v = both(fo, z1o, z2o, ed2, edb2)
    assert v[0] == MALFORMED


@pytest.mark.gpu
@pytest.mark.parametrize("n_bits,shared,K", [(1024, False, 3), (2048, True, 2), (1024, True, 1)])
def test_gpu_correct_message_matches_oracle(ctx, oracle, n_bits, shared, K):
    kw = n_bits // 32
    if n_bits == 2048:
        pq = [H.fixture_key()]
    else:
        pq = [H.test_key(n_bits, tag=t) for t in range(1 if shared else 3)]
    keys = [k[2] for k in pq]
    B = 5
    oracle.set_threads(min(8, oracle.max_threads()))
    rows, a = cm_cases(n_bits, keys, B, K, b"cm-gpu-%d-%d" % (n_bits, K))
    # proof 3: the message is not in the list (the reference panics); proof 4: r = p, so u^e has no inverse
    a["msg"][3] = L.int_to_limbs(rows[3]["valid"][0] + 1, kw)
    a["r"][4] = L.int_to_limbs(pq[4 % len(pq)][0], kw)
    n_arr = a["n"][:1] if shared else a["n"]
    stride = 0 if shared else kw
    cto, evo, zvo, avo, so = oracle.correct_message_prove(n_bits, K, n_arr, stride, a["valid"], a["msg"], a["r"], a["e_sim"], a["z_sim"], a["w"])
    ctg = np.full((B, 2 * kw), 7, np.uint32); evg = np.full((B, K, 8), 7, np.uint32); zvg = np.full((B, K, kw), 7, np.uint32)
    avg = np.full((B, K, 2 * kw), 7, np.uint32); sg = np.full(B, 9, np.uint8)
    ctx.correct_message_prove(n_bits, B, K, n_arr, stride, a["valid"], a["msg"], a["r"], a["e_sim"], a["z_sim"], a["w"], ctg, evg, zvg, avg, sg)
    assert np.array_equal(so, sg), (so, sg)
    assert list(so[:4]) == [0, 0, 0, MALFORMED] and (so[4] == MALFORMED) == (K > 1)
    for name, x, y in (("ciphertext", cto, ctg), ("e_vec", evo, evg), ("z_vec", zvo, zvg), ("a_vec", avo, avg)):
        assert np.array_equal(x, y), name

    def both(ct, ev, zv, av):
        vo = oracle.correct_message_verify(n_bits, K, n_arr, stride, a["valid"], ct, ev, zv, av)
        vg = np.full(B, 9, np.uint8)
        ctx.correct_message_verify(n_bits, B, K, n_arr, stride, a["valid"], ct, ev, zv, av, vg)
        assert np.array_equal(vo, vg), (vo, vg)
        return list(vo)
    v = both(cto, evo, zvo, avo)
    assert v[:3] == [1, 1, 1]
    zv2 = zvo.copy(); zv2[0, K - 1, 0] ^= 1
    ev2 = evo.copy(); ev2[1, 0, 7] ^= 0x80000000
    av2 = avo.copy(); av2[2, 0, 3] ^= 2
    v = both(cto, ev2, zv2, av2)
    assert v[0] == 0 and v[1] == MALFORMED and v[2] == MALFORMED      # a_vec feeds the challenge: the sum check fails first
    ct2 = cto.copy(); ct2[0, 0] ^= 1
    assert both(ct2, evo, zvo, avo)[0] == 0
