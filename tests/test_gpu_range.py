"""GPU parity tests: RangeProofNi prove / verify through the C ABI, byte-exact against the
C/GMP oracle on the same seeded inputs (behaviours of range_proof_ni.rs:148-199 and
range_proof.rs:431-525 plus tampering cases the reference has no test for)."""
import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

pytestmark = pytest.mark.gpu

OUT_FIELDS = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")


def gpu_prove(ctx, pb, wt):
    B = pb.batch
    e = np.zeros((B, 32), np.uint8); elen = np.zeros(B, np.uint8); st = np.full(B, 9, np.uint8)
    ctx.range_ni_prove(pb.struct(), wt.struct(), e, elen, st, device=False)
    return e, elen, st


def oracle_prove(oracle, pb, wt):
    B = pb.batch
    e = np.zeros((B, 32), np.uint8); elen = np.zeros(B, np.uint8); st = np.full(B, 9, np.uint8)
    oracle.range_ni_prove(pb.struct(), wt.struct(), e, elen, st)
    return e, elen, st


def clone_inputs(pb):
    q = zkp.RangeBatch(pb.n_bits, pb.batch, pb.ef, shared_key=pb.shared_key)
    q.n[:] = pb.n; q.range[:] = pb.range; q.ciphertext[:] = pb.ciphertext
    return q


def assert_same_proofs(a, b):
    for f in OUT_FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("n_bits,key_bits,batch,shared", [(1024, 512, 5, True), (1024, 1024, 3, False), (2048, 2048, 2, True)])
def test_prove_and_verify_match_oracle(ctx, oracle, n_bits, key_bits, batch, shared):
    if key_bits == 2048:
        keys = [H.fixture_key()[2]]
    else:
        keys = [H.test_key(key_bits, tag=i)[2] for i in range(1 if shared else batch)]
    cases = H.build_range_case(b"gpu-range-%d-%d" % (n_bits, key_bits), keys, n_bits, batch, shared=shared)
    # last proof of the batch is dishonest: x far outside the range (range_proof_ni.rs:180-199)
    bad = H.build_range_case(b"gpu-range-bad", [cases[-1]["n"]], n_bits, 1, honest=False)[0]
    cases[-1] = bad
    pb_o, wt = H.fill_batch(cases, n_bits, shared, oracle)
    pb_g = clone_inputs(pb_o)
    eo = oracle_prove(oracle, pb_o, wt)
    eg = gpu_prove(ctx, pb_g, wt)
    for a, b in zip(eo, eg):
        assert np.array_equal(a, b)
    assert_same_proofs(pb_o, pb_g)
    vo = np.zeros(batch, np.uint8); vg = np.full(batch, 7, np.uint8)
    oracle.range_ni_verify(pb_o.struct(), vo)
    ctx.range_ni_verify(pb_g.struct(), vg, device=False)
    assert np.array_equal(vo, vg)
    assert list(vo) == [zkp.VERDICT_ACCEPT] * (batch - 1) + [zkp.VERDICT_REJECT]


def test_verify_tampering_matches_oracle(ctx, oracle):
    n_bits = 1024
    n = H.test_key(1024)[2]
    base_cases = H.build_range_case(b"gpu-tamper", [n], n_bits, 1)
    pb0, wt = H.fill_batch(base_cases, n_bits, True, oracle)
    oracle_prove(oracle, pb0, wt)
    mask_rows = [i for i in range(128) if pb0.resp_kind[0, i] == zkp.RESP_MASK]
    open_rows = [i for i in range(128) if pb0.resp_kind[0, i] == zkp.RESP_OPEN]
    j2 = [i for i in mask_rows if pb0.resp_j[0, i] == 2]
    j1 = [i for i in mask_rows if pb0.resp_j[0, i] == 1]
    kw = n_bits // 32
    T = base_cases[0]["range"] // 3
    nn = n * n

    def t_none(p): pass
    def t_masked_r(p): p.resp_r1[0, mask_rows[0], 0] ^= 1
    def t_masked_x(p): p.resp_w1[0, mask_rows[1], 0] ^= 1
    def t_open_w2(p): p.resp_w2[0, open_rows[0], 0] ^= 1
    def t_open_r1(p): p.resp_r1[0, open_rows[1], 3] ^= 0x10
    def t_kind(p): p.resp_kind[0, open_rows[2]] = zkp.RESP_MASK
    def t_kind2(p): p.resp_kind[0, mask_rows[2]] = zkp.RESP_OPEN
    def t_c1(p): p.c1[0, open_rows[3], 5] ^= 1
    def t_c2_unused_half(p):   # c2 of a j=1 Mask row is not checked by the verifier but IS hashed -> challenge changes
        p.c2[0, j1[0], 0] ^= 1
    def t_j_other(p):          # any j != 1 selects c2 (range_proof.rs:324-328): still accepted
        p.resp_j[0, j2[0]] = 7
    def t_j_flip(p): p.resp_j[0, j2[0]] = 1
    def t_cipher(p): p.ciphertext[0, 0] ^= 1
    def t_range(p): p.range[0, 0] += 3      # moves T by one: boundary rows may flip, hash unchanged
    def t_c1_plus_nn(p):       # non-canonical c1 = c1 + n^2 on an Open row never equals a residue; the hash changes too
        i = open_rows[4]
        v = L.limbs_to_int(p.c1[0, i]) + nn
        if v.bit_length() <= 64 * kw:
            p.c1[0, i] = L.int_to_limbs(v, 2 * kw)
    def t_r_plus_n(p):         # masked_r + n: Enc uses r^n mod n^2, (r+n)^n == r^n mod n^2 -> still accepted
        i = mask_rows[3]
        v = L.limbs_to_int(p.resp_r1[0, i]) + n
        if v.bit_length() <= 32 * kw:
            p.resp_r1[0, i] = L.int_to_limbs(v, kw)
    def t_w_boundary(p):       # Open row with w2 := T exactly -> flag false (strict comparisons :300-305)
        i = open_rows[5]
        p.resp_w2[0, i] = L.int_to_limbs(T, kw)

    tampers = [t_none, t_masked_r, t_masked_x, t_open_w2, t_open_r1, t_kind, t_kind2, t_c1, t_c2_unused_half, t_j_other, t_j_flip,
               t_cipher, t_range, t_c1_plus_nn, t_r_plus_n, t_w_boundary]
    B = len(tampers)
    pb = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
    pb.n[:] = pb0.n
    for b, t in enumerate(tampers):
        for f in ("range", "ciphertext") + OUT_FIELDS:
            getattr(pb, f)[b] = getattr(pb0, f)[0]
        one = pb.slice(b, b + 1)
        t(one)
    vo = np.zeros(B, np.uint8); vg = np.full(B, 7, np.uint8)
    oracle.range_ni_verify(pb.struct(), vo)
    ctx.range_ni_verify(pb.struct(), vg, device=False)
    assert np.array_equal(vo, vg), (list(vo), list(vg))
    assert vo[0] == zkp.VERDICT_ACCEPT and vo[1] == zkp.VERDICT_REJECT
    assert vo[tampers.index(t_j_other)] == zkp.VERDICT_ACCEPT
    assert vo[tampers.index(t_r_plus_n)] == zkp.VERDICT_ACCEPT
    # the same 16 proofs tiled to 640: a call of this size takes the one-stream sequence (hash -> plan -> k_enc), the 16-proof
    # call above the two-stream one (hash next to k_enc, zkp_api_proofs.inc:range_verify_impl); both must say what the oracle says
    tiles = 40
    big = zkp.RangeBatch(n_bits, B * tiles, 128, shared_key=True)
    big.n[:] = pb.n
    for f in ("range", "ciphertext") + OUT_FIELDS:
        getattr(big, f)[:] = np.tile(getattr(pb, f), (tiles,) + (1,) * (getattr(pb, f).ndim - 1))
    vb = np.full(B * tiles, 7, np.uint8)
    ctx.range_ni_verify(big.struct(), vb, device=False)
    assert np.array_equal(vb, np.tile(vo, tiles))


def test_boundary_masked_x(ctx, oracle):
    """Mask rows accept T <= masked_x <= 2T inclusively (range_proof.rs:338); build rows at T-1, T, 2T, 2T+1
    with consistent ciphertexts so that only the range predicate decides."""
    n_bits, kw = 1024, 32
    n = H.test_key(1024)[2]
    cases = H.build_range_case(b"gpu-boundary", [n], n_bits, 1)
    pb0, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle_prove(oracle, pb0, wt)
    T = cases[0]["range"] // 3
    x, r = cases[0]["x"], cases[0]["r"]
    row = [i for i in range(128) if pb0.resp_kind[0, i] == zkp.RESP_MASK][0]
    j = int(pb0.resp_j[0, row])
    wname, rname = ("w1", "r1") if j == 1 else ("w2", "r2")
    targets = [T - 1, T, 2 * T, 2 * T + 1]
    B = len(targets)
    pb = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
    pb.n[:] = pb0.n
    for b, mx in enumerate(targets):
        for f in ("range", "ciphertext") + OUT_FIELDS:
            getattr(pb, f)[b] = getattr(pb0, f)[0]
        # replace the commitment of that row by Enc(mx - x, r_j) so that Enc(mx, r*r_j) == c_j * cipher_x holds
        rj = cases[0][rname][row]
        cj = pm.enc(n, mx - x, rj)
        (pb.c1 if j == 1 else pb.c2)[b, row] = L.int_to_limbs(cj, 2 * kw)
        pb.resp_w1[b, row] = L.int_to_limbs(mx, kw)
    # the commitment changed -> the challenge changes -> most proofs now fail on bit/kind mismatches; the oracle
    # decides what is right, the GPU must agree on every one of them
    vo = np.zeros(B, np.uint8); vg = np.full(B, 7, np.uint8)
    oracle.range_ni_verify(pb.struct(), vo)
    ctx.range_ni_verify(pb.struct(), vg, device=False)
    assert np.array_equal(vo, vg)


def test_masked_x_exactly_T_is_accepted(ctx, oracle):
    """N3: the verifier's Mask check is inclusive (range_proof.rs:338) while the prover's choice is strict (:233-234).
    With w2 = T - x on every row, x + w1 = 2T fails the strict test, the prover answers j = 2 with masked_x = T exactly,
    and the verifier must accept it."""
    n_bits, kw = 1024, 32
    n = H.test_key(1024)[2]
    cases = H.build_range_case(b"gpu-eqT", [n], n_bits, 2)
    for c in cases:
        T = c["range"] // 3
        c["w2"] = [T - c["x"]] * 128
        c["w1"] = [2 * T - c["x"]] * 128
    pb_o, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb_g = clone_inputs(pb_o)
    oracle_prove(oracle, pb_o, wt)
    gpu_prove(ctx, pb_g, wt)
    assert_same_proofs(pb_o, pb_g)
    mask = pb_g.resp_kind == zkp.RESP_MASK
    assert mask.any() and (pb_g.resp_j[mask] == 2).all()
    T0 = cases[0]["range"] // 3
    assert all(L.limbs_to_int(pb_g.resp_w1[0, i]) == T0 for i in range(128) if mask[0, i])
    vo = np.zeros(2, np.uint8); vg = np.full(2, 7, np.uint8)
    oracle.range_ni_verify(pb_o.struct(), vo)
    ctx.range_ni_verify(pb_g.struct(), vg, device=False)
    assert list(vo) == list(vg) == [zkp.VERDICT_ACCEPT] * 2


def test_device_pointer_mode_and_error_factor_zero(ctx, oracle):
    torch = pytest.importorskip("torch")
    n_bits = 1024
    n = H.test_key(512)[2]
    cases = H.build_range_case(b"gpu-devptr", [n], n_bits, 3)
    pb_o, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb_g = clone_inputs(pb_o)
    oracle_prove(oracle, pb_o, wt)
    dpb, dwt = pb_g.to("cuda"), wt.to("cuda")
    torch.cuda.synchronize()
    ctx.range_ni_prove(dpb.struct(), dwt.struct(), None, None, None, device=True)
    dv = torch.zeros(3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.range_ni_verify(dpb.struct(), dv, device=True)
    ctx.synchronize()
    back = dpb.to(None)
    assert_same_proofs(pb_o, back)
    assert dv.cpu().tolist() == [zkp.VERDICT_ACCEPT] * 3
    # error_factor = 0: all() over an empty list accepts (range_proof.rs:350)
    z = zkp.RangeBatch(n_bits, 2, 0, shared_key=True)
    z.n[:] = pb_o.n
    v = np.full(2, 7, np.uint8)
    ctx.range_ni_verify(z.struct(), v, device=False)
    vo = np.full(2, 7, np.uint8)
    oracle.range_ni_verify(z.struct(), vo)
    assert list(v) == list(vo) == [zkp.VERDICT_ACCEPT] * 2
