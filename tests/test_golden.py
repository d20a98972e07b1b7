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
