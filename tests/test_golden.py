"""Golden vectors (tests/golden/*.json, minted by tests/golden/make_golden.py) replayed against the oracle
(CPU, default run) and against the HIP engine (-m gpu)."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def iv(s):
    return int(s, 16)


# ------------------------------------------------------------------ helpers shared by both back ends
def modexp_case_arrays(bits):
    ks = [k for k in load("modexp_kat") if k["bits"] == bits]
    nl = bits // 32
    return (ks, *(L.ints_to_limbs([iv(k[f]) for k in ks], nl) for f in ("base", "exp", "mod")))


def range_case(t):
    n = H.fixture_key()[2]
    cases = H.build_range_case(t["seed"].encode(), [n], 2048, 1, honest=t["honest"])
    assert format(cases[0]["range"], "x") == t["range"] and format(cases[0]["x"], "x") == t["x"]
    return cases


def check_range_outputs(t, pb, e, elen, verdict):
    assert bytes(e[0, :elen[0]]).hex() == t["e"]
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert sha(getattr(pb, f)) == t["sha_" + f], f
    assert int(verdict[0]) == t["verdict"]


# ------------------------------------------------------------------ CPU: oracle and python model vs goldens
@pytest.mark.parametrize("bits", [2048, 4096, 8192])
def test_oracle_modexp_kats(oracle, bits):
    ks, b, e, m = modexp_case_arrays(bits)
    nl = bits // 32
    out = L.limbs_to_ints(oracle.modexp(bits, bits, b, e, nl, m, nl))
    assert out == [iv(k["out"]) for k in ks]


def test_oracle_enc_kats(oracle):
    g = load("enc_kat")
    n = iv(g["n"])
    ms = [iv(i["m"]) for i in g["items"]]; rs = [iv(i["r"]) for i in g["items"]]
    out = L.limbs_to_ints(oracle.paillier_enc(2048, L.ints_to_limbs([n], 64), 0, L.ints_to_limbs(ms, 64), L.ints_to_limbs(rs, 64)))
    assert out == [iv(i["c"]) for i in g["items"]]
    assert out[0] == pm.enc(n, ms[0], rs[0])


def test_digest_kats_python_and_oracle(oracle):
    g = load("digest_kat")
    for k in g["compute_digest"]:
        assert format(pm.compute_digest([iv(v) for v in k["items"]]), "x") == k["digest"]
    z = g["leading_zero_challenge"]
    n, c1, c2 = iv(z["n"]), [iv(v) for v in z["c1"]], [iv(v) for v in z["c2"]]
    e = oracle.fs_challenge(1024, len(c1), L.int_to_limbs(n, 32), L.ints_to_limbs(c1, 64), L.ints_to_limbs(c2, 64))
    assert e.hex() == z["e"] and len(e) < 32 and pm.fs_challenge(n, c1, c2) == e


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_oracle_range_transcripts(oracle, idx):
    t = load("range_ni_transcripts")[idx]
    oracle.set_threads(min(8, oracle.max_threads()))
    pb, wt = H.fill_batch(range_case(t), 2048, True, oracle)
    assert format(L.limbs_to_int(pb.ciphertext[0]), "x") == t["ciphertext"]
    e = np.zeros((1, 32), np.uint8); elen = np.zeros(1, np.uint8); st = np.zeros(1, np.uint8)
    oracle.range_ni_prove(pb.struct(), wt.struct(), e, elen, st)
    v = np.zeros(1, np.uint8)
    oracle.range_ni_verify(pb.struct(), v)
    check_range_outputs(t, pb, e, elen, v)
    assert (v[0] == zkp.VERDICT_ACCEPT) == t["honest"]


def test_oracle_correct_key_goldens(oracle):
    for g in load("correct_key_ni"):
        salt = bytes.fromhex(g["salt"]); n = iv(g["n"])
        rho = oracle.correct_key_rho(2048, L.int_to_limbs(n, 64), salt)
        assert format(L.limbs_to_int(rho[0]), "x") == g["rho0"] and sha(rho) == g["sha_rho"]
        p, q, _ = H.fixture_key()
        nl, sg = oracle.correct_key_ni_prove(2048, L.int_to_limbs(p, 32), L.int_to_limbs(q, 32), salt)
        assert sha(sg) == g["sha_sigma"] and format(L.limbs_to_int(sg[0]), "x") == g["sigma0"]
        assert int(oracle.correct_key_ni_verify(2048, nl[None], sg[None], salt)[0]) == g["verdict"] == zkp.VERDICT_ACCEPT
    # the survey's independent throw-away model saw rho_0 = 0x4fc11babe0c953eafcee1ad81b9a9d71... for (fixture n, "KZen")
    assert load("correct_key_ni")[0]["rho0"].startswith("4fc11babe0c953eafcee1ad81b9a9d71")


def test_oracle_dlog_goldens(oracle):
    for g in load("dlog"):
        N, gg, ni, s, r = (iv(g[k]) for k in ("N", "g", "ni", "secret", "r"))
        x, y = oracle.dlog_prove(2048, 768, *(L.ints_to_limbs([v], 64) for v in (N, gg, ni)), L.ints_to_limbs([s], 8), L.ints_to_limbs([r], 16))
        assert format(L.limbs_to_int(x[0]), "x") == g["x"] and format(L.limbs_to_int(y[0]), "x") == g["y"]
        v = oracle.dlog_verify(2048, 768, *(L.ints_to_limbs([v], 64) for v in (N, gg, ni)), x, y)
        assert int(v[0]) == g["verdict"]


# ------------------------------------------------------------------ GPU: the engine vs goldens
@pytest.mark.gpu
@pytest.mark.parametrize("bits", [2048, 4096, 8192])
def test_gpu_modexp_kats(ctx, bits):
    ks, b, e, m = modexp_case_arrays(bits)
    nl = bits // 32
    out = np.zeros_like(b)
    ctx.modexp(bits, bits, len(ks), b, e, nl, m, nl, out)
    assert L.limbs_to_ints(out) == [iv(k["out"]) for k in ks]


@pytest.mark.gpu
def test_gpu_enc_kats(ctx):
    g = load("enc_kat")
    n = iv(g["n"])
    ms = [iv(i["m"]) for i in g["items"]]; rs = [iv(i["r"]) for i in g["items"]]
    out = np.zeros((len(ms), 128), np.uint32)
    ctx.paillier_enc(2048, len(ms), L.ints_to_limbs([n], 64), 0, L.ints_to_limbs(ms, 64), L.ints_to_limbs(rs, 64), out)
    assert L.limbs_to_ints(out) == [iv(i["c"]) for i in g["items"]]


@pytest.mark.gpu
def test_gpu_range_transcripts(ctx, oracle):
    for t in load("range_ni_transcripts"):
        pb, wt = H.fill_batch(range_case(t), 2048, True, oracle)     # oracle only computes the input ciphertext Enc(x, r)
        assert format(L.limbs_to_int(pb.ciphertext[0]), "x") == t["ciphertext"]
        e = np.zeros((1, 32), np.uint8); elen = np.zeros(1, np.uint8); st = np.zeros(1, np.uint8)
        ctx.range_ni_prove(pb.struct(), wt.struct(), e, elen, st, device=False)
        v = np.full(1, 9, np.uint8)
        ctx.range_ni_verify(pb.struct(), v, device=False)
        check_range_outputs(t, pb, e, elen, v)


@pytest.mark.gpu
def test_gpu_leading_zero_challenge(ctx, oracle):
    """N2 on the device hash: a transcript whose digest starts with a 00 byte"""
    z = load("digest_kat")["leading_zero_challenge"]
    n, c1, c2 = iv(z["n"]), [iv(v) for v in z["c1"]], [iv(v) for v in z["c2"]]
    pb = zkp.RangeBatch(1024, 1, 2, shared_key=True)
    pb.n[0] = L.int_to_limbs(n, 32)
    pb.c1[0] = L.ints_to_limbs(c1, 64); pb.c2[0] = L.ints_to_limbs(c2, 64)
    pb.range[0, 0] = 1000
    vg = np.full(1, 9, np.uint8); vo = np.full(1, 9, np.uint8)
    ctx.range_ni_verify(pb.struct(), vg, device=False)
    oracle.range_ni_verify(pb.struct(), vo)
    assert vg[0] == vo[0]     # rows are garbage: what matters is that both derive the same bits from the 31-byte challenge
    # flip response kinds so that the verdict depends on every one of the first two challenge bits
    e = bytes.fromhex(z["e"])
    for trial in range(4):
        pb.resp_kind[0, 0] = trial & 1; pb.resp_kind[0, 1] = trial >> 1
        ctx.range_ni_verify(pb.struct(), vg, device=False)
        oracle.range_ni_verify(pb.struct(), vo)
        assert vg[0] == vo[0]


@pytest.mark.gpu
def test_gpu_correct_key_goldens(ctx, oracle):
    p, q, n = H.fixture_key()
    for g in load("correct_key_ni"):
        salt = bytes.fromhex(g["salt"])
        nl, sg = oracle.correct_key_ni_prove(2048, L.int_to_limbs(p, 32), L.int_to_limbs(q, 32), salt)   # prover side stays on the CPU
        assert sha(sg) == g["sha_sigma"]
        bad = sg.copy(); bad[5, 0] ^= 1
        v = np.full(2, 9, np.uint8)
        ctx.correct_key_ni_verify(2048, 2, np.stack([nl, nl]), np.stack([sg, bad]), salt, v)
        assert list(v) == [g["verdict"], g["verdict_tampered_sigma5"]]


# ------------------------------------------------------------------ SURVEY 8(f) rows (next_rows.json): one replay, two back ends
class _OracleBackend:
    """adapter: the same calls as zkp.Context, answered by the oracle"""
    def __init__(self, o): self.o = o
    def modinv(self, bits, cnt, a, m, ms, out, st): o, s = self.o.modinv(bits, a, m, ms); out[:] = o; st[:] = s
    def mul_proof_prove(self, nb, B, n, ns, e_a, e_b, e_c, a, b, r_a, r_b, r_c, d, r_d, f, z1, z2, e_d, e_db, st):
        r = self.o.mul_proof_prove(nb, n, ns, e_a, e_b, e_c, a, b, r_a, r_b, r_c, d, r_d)
        for dst, src in zip((f, z1, z2, e_d, e_db, st), r): dst[:] = src
    def mul_proof_verify(self, nb, B, n, ns, e_a, e_b, e_c, f, z1, z2, e_d, e_db, v): v[:] = self.o.mul_proof_verify(nb, n, ns, e_a, e_b, e_c, f, z1, z2, e_d, e_db)
    def correct_message_prove(self, nb, B, K, n, ns, valid, msg, r, es, zs, w, ct, ev, zv, av, st):
        res = self.o.correct_message_prove(nb, K, n, ns, valid, msg, r, es, zs, w)
        for dst, src in zip((ct, ev, zv, av, st), res): dst[:] = src
    def correct_message_verify(self, nb, B, K, n, ns, valid, ct, ev, zv, av, v): v[:] = self.o.correct_message_verify(nb, K, n, ns, valid, ct, ev, zv, av)
    def zero_proof_prove(self, nb, B, n, ns, c, r, rp, z, a): zz, aa = self.o.zero_proof_prove(nb, n, ns, c, r, rp); z[:] = zz; a[:] = aa
    def ciphertext_proof_prove(self, nb, B, n, ns, c, x, r, xp, rp, z1, z2, cp):
        a, b, c_ = self.o.ciphertext_proof_prove(nb, n, ns, c, x, r, xp, rp); z1[:] = a; z2[:] = b; cp[:] = c_
    def decimal_to_limbs(self, text, items, dst, st): self.o.decimal_to_limbs(text, items, dst, st)
    def limbs_to_decimal(self, src): return self.o.limbs_to_decimal(src, 32 * src.shape[1] * 30103 // 100000 + 2)


def replay_next_rows(be):
    g = load("next_rows")
    n = iv(g["n"]); kw = 32
    A = lambda v, w: L.ints_to_limbs([v] if isinstance(v, int) else v, w)
    nl = A(n, kw)
    # mod_inv
    vals = [iv(k["a"]) for k in g["mod_inv"]]
    out = np.full((len(vals), 64), 7, np.uint32); st = np.full(len(vals), 9, np.uint8)
    be.modinv(2048, len(vals), A(vals, 64), A(n * n, 64), 0, out, st)
    assert [format(v, "x") for v in L.limbs_to_ints(out)] == [k["inv"] for k in g["mod_inv"]] and list(st) == [k["status"] for k in g["mod_inv"]]
    # MulProof
    for k in g["mul_proof"]:
        ins = [A(iv(k[f]), 64 if f.startswith("e_") else kw) for f in ("e_a", "e_b", "e_c", "a", "b", "r_a", "r_b", "r_c", "d", "r_d")]
        f = np.zeros((1, kw), np.uint32); z1, z2, e_d, e_db = (np.zeros((1, 64), np.uint32) for _ in range(4)); s1 = np.full(1, 9, np.uint8)
        be.mul_proof_prove(1024, 1, nl, 0, *ins, f, z1, z2, e_d, e_db, s1)
        assert [format(L.limbs_to_int(x[0]), "x") for x in (f, z1, z2, e_d, e_db)] == [k[q] for q in ("f", "z1", "z2", "e_d", "e_db")] and s1[0] == 0
        v = np.full(1, 9, np.uint8)
        be.mul_proof_verify(1024, 1, nl, 0, ins[0], ins[1], ins[2], f, z1, z2, e_d, e_db, v)
        assert int(v[0]) == k["verdict"]
    # CorrectMessageProof
    c = g["correct_message"]
    K = len(c["valid"])
    ct = np.zeros((1, 64), np.uint32); ev = np.zeros((1, K, 8), np.uint32); zv = np.zeros((1, K, kw), np.uint32); av = np.zeros((1, K, 64), np.uint32)
    s1 = np.full(1, 9, np.uint8)
    valid = A(c["valid"], kw)[None]
    be.correct_message_prove(1024, 1, K, nl, 0, valid, A(c["message"], kw), A(iv(c["r"]), kw), A([iv(v) for v in c["e_sim"]], 8)[None],
                             A([iv(v) for v in c["z_sim"]], kw)[None], A(iv(c["w"]), kw), ct, ev, zv, av, s1)
    assert format(L.limbs_to_int(ct[0]), "x") == c["ciphertext"] and s1[0] == 0
    for arr, key in ((ev, "e_vec"), (zv, "z_vec"), (av, "a_vec")):
        assert [format(v, "x") for v in L.limbs_to_ints(arr[0])] == c[key], key
    v = np.full(1, 9, np.uint8)
    be.correct_message_verify(1024, 1, K, nl, 0, valid, ct, ev, zv, av, v)
    assert int(v[0]) == c["verdict"]
    # ZeroProof / CiphertextProof provers
    s = g["sigma"]
    z = np.zeros((1, 64), np.uint32); a_ = np.zeros((1, 64), np.uint32)
    be.zero_proof_prove(1024, 1, nl, 0, A(iv(s["c0"]), 64), A(iv(s["r"]), kw), A(iv(s["r_prime"]), kw), z, a_)
    assert format(L.limbs_to_int(z[0]), "x") == s["zero_z"] and format(L.limbs_to_int(a_[0]), "x") == s["zero_a"]
    z1 = np.zeros((1, kw + 16), np.uint32); z2 = np.zeros((1, 64), np.uint32); cp = np.zeros((1, 64), np.uint32)
    be.ciphertext_proof_prove(1024, 1, nl, 0, A(iv(s["cx"]), 64), A(iv(s["x"]), kw), A(iv(s["r"]), kw), A(iv(s["x_prime"]), kw), A(iv(s["r_prime"]), kw), z1, z2, cp)
    assert [format(L.limbs_to_int(q[0]), "x") for q in (z1, z2, cp)] == [s["ct_z1"], s["ct_z2"], s["ct_c_prime"]]
    # decimal wire format, both directions
    texts = [k["text"].encode() for k in g["decimal"]]
    blob = b",".join(texts)
    items = (zkp.DecItem * len(texts))()
    pos = 0
    for i, t in enumerate(texts):
        items[i].text_off = pos; items[i].len = len(t); items[i].words = 64; items[i].dst_off = i * 64
        pos += len(t) + 1
    dst = np.full((len(texts), 64), 7, np.uint32); st = np.full(len(texts), 9, np.uint8)
    be.decimal_to_limbs(blob, items, dst, st)
    assert [format(v, "x") for v in L.limbs_to_ints(dst)] == [k["hex"] for k in g["decimal"]] and not st.any()
    assert be.limbs_to_decimal(dst) == texts


def test_oracle_next_rows_goldens(oracle):
    replay_next_rows(_OracleBackend(oracle))


@pytest.mark.gpu
def test_gpu_next_rows_goldens(ctx):
    replay_next_rows(ctx)
