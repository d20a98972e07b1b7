"""BASELINE.json full-size configurations on the GPU, checked through size-independent properties
(prove -> verify round trips, tamper -> reject, the Paillier homomorphism) plus oracle parity on samples."""
import importlib

import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

pytestmark = pytest.mark.gpu
synth = importlib.import_module("zk-paillier_amd.synth")


@pytest.fixture(autouse=True)
def _throughput_engine_only(ctx):
    if ctx.test_geometry != 36:
        pytest.skip("full-size batches belong to the throughput engine (the latency engine serves calls of a few proofs)")


def test_config2_3_batch4096_prove_verify_n2048(ctx, oracle):
    torch = pytest.importorskip("torch")
    B, n_bits = 4096, 2048
    dev = torch.device("cuda", 0)
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, n_bits, B, seed=77, device=dev)
    torch.cuda.synchronize()
    ctx.paillier_enc(n_bits, B, pb.n, 0, wt.x, wt.r, pb.ciphertext)
    status = torch.full((B,), 9, dtype=torch.uint8, device=dev)
    ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, status, device=True)      # config 3
    ctx.synchronize()
    assert int(status.sum()) == 0
    # tamper three proofs in different ways
    pb.resp_r1[5, 0, 0] ^= 1
    pb.c1[100, 17, 3] ^= 4
    pb.resp_kind[2000, 64] ^= 1
    torch.cuda.synchronize()
    verdict = torch.full((B,), 9, dtype=torch.uint8, device=dev)
    ctx.range_ni_verify(pb.struct(), verdict, device=True)                              # config 2
    ctx.synchronize()
    v = verdict.cpu().numpy()
    expect = np.ones(B, np.uint8); expect[[5, 100, 2000]] = 0
    assert np.array_equal(v, expect)
    # oracle parity on a sample of the batch (tampered ones included)
    idx = [0, 5, 100, 2000, 4095]
    host = pb.to(None)
    sample = zkp.RangeBatch(n_bits, len(idx), 128, shared_key=True)
    sample.n[:] = host.n
    for k, b in enumerate(idx):
        for f in ("range", "ciphertext", "c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
            getattr(sample, f)[k] = getattr(host, f)[b]
    oracle.set_threads(min(16, oracle.max_threads()))
    vo = np.zeros(len(idx), np.uint8)
    oracle.range_ni_verify(sample.struct(), vo)
    assert list(vo) == [int(v[b]) for b in idx]
    # the prove output itself against the oracle for one untouched proof
    hw = wt.to(None)
    one = zkp.RangeBatch(n_bits, 1, 128, shared_key=True)
    one.n[:] = host.n; one.range[0] = host.range[4095]
    w1 = zkp.make_range_witness(n_bits, 1)
    for f in ("x", "r", "w1", "w2", "r1", "r2"):
        getattr(w1, f)[0] = getattr(hw, f)[4095]
    oracle.range_ni_prove(one.struct(), w1.struct(), None, None, None)
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(one, f)[0], getattr(host, f)[4095]), f


def test_paillier_homomorphism_large_batch(ctx):
    """Enc(m1,r1) * Enc(m2,r2) mod n^2 == Enc(m1+m2, r1*r2 mod n) for 20000 random pairs (no oracle involved)"""
    n = synth.BENCH_N
    kw, count = 64, 20000
    rng = np.random.default_rng(5)
    def rnd(bits):
        a = rng.integers(0, 1 << 32, size=(count, kw), dtype=np.uint64).astype(np.uint32)
        full, rem = bits // 32, bits % 32
        a[:, full] &= (1 << rem) - 1; a[:, full + 1:] = 0
        return a
    m1, m2 = rnd(250), rnd(250)
    r1, r2 = rnd(2039), rnd(2039)
    nl = L.ints_to_limbs([n], kw)
    c1 = np.zeros((count, 2 * kw), np.uint32); c2 = np.zeros_like(c1); c3 = np.zeros_like(c1); prod = np.zeros_like(c1)
    ctx.paillier_enc(2048, count, nl, 0, m1, r1, c1)
    ctx.paillier_enc(2048, count, nl, 0, m2, r2, c2)
    nn = L.ints_to_limbs([n * n], 2 * kw)
    ctx.modmul(4096, count, c1, c2, nn, 0, prod)
    r12 = np.zeros_like(r1)
    ctx.modmul(2048, count, r1, r2, nl, 0, r12)
    msum = (m1.astype(np.uint64) + 0)  # m1 + m2 with carries (values < 2^250: plain limb add with carry)
    carry = np.zeros(count, np.uint64); m12 = np.zeros_like(m1)
    for k in range(kw):
        s = m1[:, k].astype(np.uint64) + m2[:, k].astype(np.uint64) + carry
        m12[:, k] = (s & 0xFFFFFFFF).astype(np.uint32); carry = s >> 32
    ctx.paillier_enc(2048, count, nl, 0, m12, r12, c3)
    assert np.array_equal(prod, c3)
    assert L.limbs_to_int(c1[0]) == pm.enc(n, L.limbs_to_int(m1[0]), L.limbs_to_int(r1[0]))


def test_config4_batch65536_correct_key_verify(ctx, oracle):
    """65536 (key, proof) records with per-record moduli: 5 real keys cycled through the batch (generating 65536
    RSA moduli is out of proportion), every 97th record tampered, every 1013th record with a pseudo-modulus."""
    n_bits, kw, B = 2048, 64, 65536
    salt = pm.SALT_STRING
    keys = [H.fixture_key()] + [H.test_key(2048, tag=t) for t in (1, 2)]
    recs = []
    for p, q, n in keys:
        nl, sg = oracle.correct_key_ni_prove(n_bits, L.int_to_limbs(p, 32), L.int_to_limbs(q, 32), salt)
        recs.append((nl, sg))
    n_arr = np.zeros((B, kw), np.uint32); s_arr = np.zeros((B, 11, kw), np.uint32)
    for k, (nl, sg) in enumerate(recs):
        n_arr[k::len(recs)] = nl; s_arr[k::len(recs)] = sg
    expect = np.ones(B, np.uint8)
    tam = np.arange(0, B, 97); s_arr[tam, 4, 2] ^= 8; expect[tam] = 0
    rng = np.random.default_rng(9)
    pseudo = np.arange(11, B, 1013)
    n_arr[pseudo] = rng.integers(0, 1 << 32, size=(len(pseudo), kw), dtype=np.uint64).astype(np.uint32) | 1
    expect[pseudo] = 0
    v = np.full(B, 9, np.uint8)
    ctx.correct_key_ni_verify(n_bits, B, n_arr, s_arr, salt, v)
    assert np.array_equal(v, expect)
    idx = np.concatenate([tam[:3], pseudo[:3], [1, 2, 3]])
    assert np.array_equal(oracle.correct_key_ni_verify(n_bits, n_arr[idx], s_arr[idx], salt), v[idx])


def test_config5_n4096_prove_verify(ctx, oracle):
    """n = 4096 (8192-bit n^2, 32 lanes per integer): oracle parity on 1 proof, round trip on 24"""
    n_bits = 4096
    n = H.test_key(4096, tag=3)[2]
    cases = H.build_range_case(b"cfg5", [n], n_bits, 24)
    cases[23] = H.build_range_case(b"cfg5-bad", [n], n_bits, 1, honest=False)[0]
    oracle.set_threads(min(16, oracle.max_threads()))
    pb, wt = H.fill_batch(cases, n_bits, True, oracle)
    ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=False)
    v = np.full(24, 9, np.uint8)
    ctx.range_ni_verify(pb.struct(), v, device=False)
    assert list(v) == [1] * 23 + [0]
    one, w1 = pb.slice(0, 1), None
    ref = zkp.RangeBatch(n_bits, 1, 128, shared_key=True)
    ref.n[:] = pb.n; ref.range[0] = pb.range[0]; ref.ciphertext[0] = pb.ciphertext[0]
    w1 = zkp.make_range_witness(n_bits, 1)
    for f in ("x", "r", "w1", "w2", "r1", "r2"):
        getattr(w1, f)[0] = getattr(wt, f)[0]
    oracle.range_ni_prove(ref.struct(), w1.struct(), None, None, None)
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(ref, f)[0], getattr(pb, f)[0]), f
    vo = np.zeros(1, np.uint8)
    oracle.range_ni_verify(ref.struct(), vo)
    assert vo[0] == 1


def test_config5_n4096_batch512_real_key(ctx, oracle):
    """BASELINE configs[4] at a per-GPU share of an 8-GPU node: B = 512 proofs at n = 4096 under the 4096-bit key of
    bench_keys.json, device-resident; prove -> verify round trip with tampering, oracle parity on two proofs (prove output and verdict)"""
    torch = pytest.importorskip("torch")
    B, n_bits = 512, 4096
    dev = torch.device("cuda", 0)
    n = synth.bench_key_4096()[2]
    assert n.bit_length() == 4096
    pb, wt = synth.synth_range_inputs(n, n_bits, B, seed=4096, device=dev)
    torch.cuda.synchronize()
    ctx.paillier_enc(n_bits, B, pb.n, 0, wt.x, wt.r, pb.ciphertext)
    status = torch.full((B,), 9, dtype=torch.uint8, device=dev)
    ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, status, device=True)
    ctx.synchronize()
    assert int(status.sum()) == 0
    pb.resp_r2[7, 3, 1] ^= 2
    pb.c2[300, 100, 200] ^= 1
    torch.cuda.synchronize()
    verdict = torch.full((B,), 9, dtype=torch.uint8, device=dev)
    ctx.range_ni_verify(pb.struct(), verdict, device=True)
    ctx.synchronize()
    v = verdict.cpu().numpy()
    # (a tampered field only matters if the challenge bit of its row selects that kind of response: check against the oracle below)
    assert v[[0, 1, 2, 511]].tolist() == [1, 1, 1, 1] and (v == 1).sum() >= B - 2
    idx = [0, 7, 300, 511]
    host = pb.to(None)
    sample = zkp.RangeBatch(n_bits, len(idx), 128, shared_key=True)
    sample.n[:] = host.n
    for k, b in enumerate(idx):
        for f in ("range", "ciphertext", "c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
            getattr(sample, f)[k] = getattr(host, f)[b]
    oracle.set_threads(min(16, oracle.max_threads()))
    vo = np.zeros(len(idx), np.uint8)
    oracle.range_ni_verify(sample.struct(), vo)
    assert list(vo) == [int(v[b]) for b in idx]
    hw = wt.to(None)
    for b in (1, 511):                                   # the prove output itself, untouched proofs
        one = zkp.RangeBatch(n_bits, 1, 128, shared_key=True)
        one.n[:] = host.n; one.range[0] = host.range[b]
        w1 = zkp.make_range_witness(n_bits, 1)
        for f in ("x", "r", "w1", "w2", "r1", "r2"):
            getattr(w1, f)[0] = getattr(hw, f)[b]
        oracle.range_ni_prove(one.struct(), w1.struct(), None, None, None)
        for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
            assert np.array_equal(getattr(one, f)[0], getattr(host, f)[b]), (b, f)


def test_per_proof_keys_n2048_range_proof_ni(ctx, oracle):
    """SURVEY 8(d) config 3 "distinct eks": every proof under its own real 2048-bit RSA modulus (pooled primes of bench_keys.json):
    the fixed-window ladder and per-key set-up at full width, byte-exact prove transcripts and verdicts against the oracle"""
    n_bits, B = 2048, 5
    keys = synth.distinct_keys_2048(40)[17:17 + B]
    assert len(set(keys)) == B and all(k.bit_length() == 2048 for k in keys)
    cases = H.build_range_case(b"per-key-2048", keys, n_bits, B, shared=False)
    cases[2] = H.build_range_case(b"per-key-2048-bad", [keys[2]], n_bits, 1, honest=False)[0]
    oracle.set_threads(min(16, oracle.max_threads()))
    pb_o, wt = H.fill_batch(cases, n_bits, False, oracle)
    pb_g = pb_o.to(None)
    e_o = np.zeros((B, 32), np.uint8); l_o = np.zeros(B, np.uint8); e_g = np.zeros((B, 32), np.uint8); l_g = np.zeros(B, np.uint8)
    oracle.range_ni_prove(pb_o.struct(), wt.struct(), e_o, l_o, None)
    ctx.range_ni_prove(pb_g.struct(), wt.struct(), e_g, l_g, None, device=False)
    assert np.array_equal(e_o, e_g) and np.array_equal(l_o, l_g)
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(pb_o, f), getattr(pb_g, f)), f
    pb_g.resp_r1[4, 9, 0] ^= 1; pb_o.resp_r1[4, 9, 0] ^= 1
    vo = np.full(B, 9, np.uint8); vg = np.full(B, 9, np.uint8)
    oracle.range_ni_verify(pb_o.struct(), vo)
    ctx.range_ni_verify(pb_g.struct(), vg, device=False)
    assert list(vo) == list(vg) and vg[0] == 1 and vg[2] == 0
