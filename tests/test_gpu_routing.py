"""The library's OWN choice of kernels — nothing pinned, $ZKP_BASEN not set — at the batch sizes where it flips, against the oracle.

A RangeProofNi call at n = 2048 under one key is served by one of these kernel families (csrc/zkp_api.hip: route_latency, launch_basen):
  * the latency engine (9 limbs per lane, libzkp_hip_lat.so) up to 96 proofs: ONE Enc per wavefront on the five-group base-n ladder
    (k_enc_basen_r2l) up to 8 proofs, the window ladder on the n^2-sized product up to 16, 8 Enc per wavefront in base-n form
    (k_enc_basen<8>) from there on — except
  * 41 ... 64 and 129 ... 192 proofs: the mid engine (18 limbs per lane, libzkp_hip_mid.so), 16 Enc per wavefront in base-n form,
  * 65 ... 96 proofs: TWO concurrent calls — up to 64 proofs on the mid engine, at least 16 on a second ctx of the latency engine —,
    and 17 ... 20 proofs: 16 on the latency engine's window ladder, the rest beside them on its one-Enc-per-wavefront ladder
    (csrc/zkp_api_proofs.inc: range_split_plan / range_split_run; zkp_diag_last_split),
  * the throughput engine's base-n kernels (k_enc_basen<2>, 32 Enc per wavefront) beyond,
  * its n^2-sized kernels (k_enc<4, .>) for keys the form does not take and for launches pinned to that engine that leave SIMDs idle.
The parity suites pin each family in turn (tests/conftest.py: ctx); this file lets the library choose, says which family it expects for
every size (zkp_ctx_last_geometry, zkp_diag_basen_last), and checks prove transcripts and verdict vectors against the C/GMP oracle on
samples of every batch — and against the OTHER form's bytes for the whole batch."""
import os

import numpy as np
import pytest

import helpers as H
from helpers import L

zkp = H.zkp
pytestmark = pytest.mark.gpu

FIELDS = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")


@pytest.fixture(scope="module")
def actx():
    assert "ZKP_BASEN" not in os.environ, "this file tests the library's default routing: unset ZKP_BASEN"
    c = zkp.Context(0)
    assert c.enc_form() == zkp.capi.ENC_FORM_AUTO
    yield c
    c.close()


def compute_units():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count


def expected_family(c, items, listed=False):
    """what csrc/zkp_api.hip is expected to pick for a Paillier launch of `items` Enc under ONE 2048-bit key that the base-n form takes
    (route_latency with one_key_paillier, launch_basen of either engine).  listed: `items` is the bound of a verify's work list — three
    quarters of it (+ 3 %) are expected to exist, and the mid engine's one-round window is judged by that (csrc/zkp_api.hip expected_items)"""
    lat, mid = c.latency_limbs_per_lane(), c.mid_limbs_per_lane()
    simds = 4 * compute_units()
    one_round = items <= 16 * simds or (listed and (3 * items + 3) // 4 + items // 32 <= 16 * simds)
    lat_round = listed and (3 * items + 3) // 4 + items // 32 <= 8 * simds      # a verify that fits one round of the latency engine's k_enc_basen<8>
    if lat == 9 and mid == 18 and ((16 * simds < items <= 24 * simds and not one_round) or 4 * simds < items <= 5 * simds or (8 * simds < items <= 9 * simds and not lat_round)
                                  or (not listed and 32 * simds < items <= 40 * simds)):
        return "split"                                         # two concurrent calls (expected_tail below)
    if mid == 18 and ((10 * simds < items and one_round) or 32 * simds < items <= 48 * simds):
        return "mid-basen"                                     # 16 Enc per wavefront: one (two) wavefronts per SIMD of the mid engine
    if lat == 9 and items <= 3 * simds * 8:                    # the latency engine: up to three wavefronts per SIMD at 8 Enc per wavefront
        if (min(items, (3 * items + 3) // 4 + items // 32) if listed else items) <= 2 * simds:
            return "lat-r2l"                                   # one Enc per wavefront, the five-group ladder (kernels_basen_r2l.hpp); a verify by its expected items (9, 10 proofs)
        return "lat-basen" if items > simds * 4 else "lat-n2"
    return "base-n" if items > simds * 16 else "n2"


def expected_tail(B):
    """proofs of a split call that run on the second latency-engine ctx (csrc/zkp_api_proofs.inc: range_split_plan; 256 Enc per proof)"""
    simds = 4 * compute_units()
    full, least = 16 * simds // 256, 4 * simds // 256
    if B * 256 <= 5 * simds:
        return B - least                                       # 17 ... 20 proofs: 16 on the window ladder, the rest on the one-Enc-per-wavefront ladder
    if B * 256 <= 9 * simds:
        return B - 8 * simds // 256                            # 33 ... 36 proofs (prove): 32 on k_enc_basen<8>, the rest on the one-Enc-per-wavefront ladder
    if B * 256 > 32 * simds:
        t = B - 32 * simds // 256                              # 129 ... 160 proofs (prove): 128 on the throughput engine, the rest beside them on the latency engine
        return 2 * simds // 256 + 1 if 3 * simds // 2 < t * 256 <= 2 * simds else t      # (not 7 or 8 proofs: the r2l ladder's workgroups would not fit beside the head)
    return B - full if B - full >= least else least            # 65 ... 96: 64 on the mid engine | at least 16 on the latency engine


def family_that_ran(c):
    if c.last_split() > 0:
        return "split"
    g = c.last_geometry()
    if g == 18:
        lanes, ok = c.diag_basen_last()
        return "mid-basen" if (lanes == 4 and ok) else "mid-n2"
    if g != zkp.load().zkp_build_limbs_per_lane():
        if c.r2l_last():
            return "lat-r2l"
        lanes, ok = c.diag_basen_last()
        return "lat-basen" if (lanes == 8 and ok) else "lat-n2"
    lanes, ok = c.diag_basen_last()
    return "base-n" if (lanes == 2 and ok) else "n2"


def sub_batch(pb, idx, n_bits):
    s = zkp.RangeBatch(n_bits, len(idx), pb.ef, shared_key=True)
    s.n[:] = pb.n
    for k, b in enumerate(idx):
        for f in ("range", "ciphertext") + FIELDS:
            getattr(s, f)[k] = getattr(pb, f)[b]
    return s


@pytest.mark.parametrize("B", [1, 2, 4, 10, 12, 18, 32, 34, 64, 65, 80, 96, 128, 136, 160, 300])
def test_default_routing_prove_and_verify_against_the_oracle(actx, oracle, B):
    n_bits, kw = 2048, 64
    n = H.fixture_key()[2]
    cases = H.build_range_case(b"routing-%d" % B, [n], n_bits, B)
    oracle.set_threads(min(16, oracle.max_threads()))
    pb_o, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
    pb.n[:] = pb_o.n; pb.range[:] = pb_o.range; pb.ciphertext[:] = pb_o.ciphertext
    actx.set_geometry(0)
    actx.set_enc_form("auto")
    status = np.full(B, 9, np.uint8)
    actx.range_ni_prove(pb.struct(), wt.struct(), None, None, status, device=False)
    want = expected_family(actx, 2 * 128 * B)
    assert family_that_ran(actx) == want, (B, family_that_ran(actx), want)
    if want == "split":
        assert actx.last_split() == expected_tail(B)
    if want == "lat-r2l":                                    # one Enc per compute unit: five wavefronts per Enc (k_enc_basen_r2l5); beyond: one wavefront per Enc
        assert actx.r2l_lanes_last() == (36 if 2 * 128 * B <= compute_units() else 12), (B, actx.r2l_lanes_last())
    assert not status.any()
    # the prove transcripts of a sample of the batch, byte for byte against the oracle
    idx = sorted({0, 1, B // 3, B // 2, B - 2, B - 1} & set(range(B)))
    so = sub_batch(pb_o, idx, n_bits)
    sw = zkp.make_range_witness(n_bits, len(idx))
    for k, b in enumerate(idx):
        for f in ("x", "r", "w1", "w2", "r1", "r2"):
            getattr(sw, f)[k] = getattr(wt, f)[b]
    oracle.range_ni_prove(so.struct(), sw.struct(), None, None, None)
    for k, b in enumerate(idx):
        for f in FIELDS:
            assert np.array_equal(getattr(so, f)[k], getattr(pb, f)[b]), (B, b, f)
    # the whole batch against other kernels: the throughput engine's n^2-sized ones (its base-n ones when those were the choice)
    other = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
    other.n[:] = pb.n; other.range[:] = pb.range; other.ciphertext[:] = pb.ciphertext
    actx.set_geometry(zkp.load().zkp_build_limbs_per_lane())
    actx.set_enc_form("basen" if want == "n2" else "n2")
    actx.range_ni_prove(other.struct(), wt.struct(), None, None, None, device=False)
    assert family_that_ran(actx) == ("base-n" if want == "n2" else "n2")
    for f in FIELDS:
        assert np.array_equal(getattr(other, f), getattr(pb, f)), (B, f)
    actx.set_geometry(0)
    actx.set_enc_form("auto")
    # verify: every 7th proof tampered in one of three ways
    bad = list(range(3, B, 7)) or [B - 1]
    for k, b in enumerate(bad):
        if k % 3 == 0:
            pb.resp_r1[b, k % 128, 0] ^= 1
        elif k % 3 == 1:
            pb.c2[b, (5 * k) % 128, 7] ^= 0x10
        else:
            pb.resp_w1[b, (11 * k) % 128, 1] ^= 2
    v = np.full(B, 9, np.uint8)
    actx.range_ni_verify(pb.struct(), v, device=False)
    want_v = expected_family(actx, 2 * 128 * B, listed=True)
    assert family_that_ran(actx) == want_v, (B, "verify", family_that_ran(actx), want_v)
    if want_v == "split":
        assert actx.last_split() == expected_tail(B)
    expect = np.ones(B, np.uint8); expect[bad] = 0
    assert np.array_equal(v, expect)
    vidx = sorted(set(idx) | set(bad[:3]) | {bad[-1]})
    vo = np.full(len(vidx), 9, np.uint8)
    sv = sub_batch(pb, vidx, n_bits)                     # (kept alive across the call: struct() holds raw pointers into its arrays)
    oracle.range_ni_verify(sv.struct(), vo)
    assert list(vo) == [int(v[b]) for b in vidx]


def test_enc_launch_threshold(actx):
    """zkp_paillier_enc_batch under one key: a launch whose n^2-sized wavefronts all find a SIMD of their own stays on those kernels,
    a larger one takes the base-n form (pinned to the throughput engine: the latency engine would take the small one)"""
    import random
    rnd = random.Random(11)
    n_bits, kw = 2048, 64
    n = rnd.getrandbits(n_bits) | 1 | (1 << (n_bits - 1))
    nw = L.int_to_limbs(n, kw)
    actx.set_geometry(zkp.load().zkp_build_limbs_per_lane())
    actx.set_enc_form("auto")
    try:
        cus = compute_units()
        for count in (200, 4 * cus * 16, 4 * cus * 16 + 32):
            mw = np.zeros((count, kw), np.uint32); mw[:, 0] = np.arange(count)
            rw = np.zeros((count, kw), np.uint32); rw[:, 0] = 3 + np.arange(count)
            out = np.zeros((count, 2 * kw), np.uint32)
            actx.paillier_enc(n_bits, count, nw, 0, mw, rw, out)
            assert family_that_ran(actx) == expected_family_throughput(cus, count), count
            for i in (0, count // 2, count - 1):
                got = sum(int(w) << (32 * j) for j, w in enumerate(out[i]))
                assert got == (1 + i * n) * pow(3 + i, n, n * n) % (n * n)
    finally:
        actx.set_geometry(0)


def expected_family_throughput(cus, items):
    return "base-n" if items > 4 * cus * 16 else "n2"


def test_the_environment_is_read_once_at_ctx_create(actx):
    """$ZKP_BASEN presets a NEW ctx and is never looked at again (round-4 verdict: routing by getenv on every launch)"""
    assert actx.enc_form() == zkp.capi.ENC_FORM_AUTO
    os.environ["ZKP_BASEN"] = "0"
    try:
        assert actx.enc_form() == zkp.capi.ENC_FORM_AUTO          # the live ctx does not change
        c2 = zkp.Context(0)
        assert c2.enc_form() == zkp.capi.ENC_FORM_N2
        os.environ["ZKP_BASEN"] = "always"
        assert c2.enc_form() == zkp.capi.ENC_FORM_N2
        c2.set_enc_form("shared")
        assert c2.enc_form() == zkp.capi.ENC_FORM_SHARED
        with pytest.raises(zkp.ZkpError):
            c2.set_enc_form(7)
        c2.close()
    finally:
        os.environ.pop("ZKP_BASEN", None)


def test_split_call_device_resident_and_switched_off(actx, oracle):
    """a call the library cuts in two (88 proofs: 64 on the mid engine, 24 beside them on the latency engine — prove AND verify: the expected items of
    an 88-proof verify do not fit the mid engine's one round; up to 81 proofs a verify is not cut, test_default_routing_…) with DEVICE-resident arrays —
    the two streams are ordered against the ctx's stream by events, not by the host —, and the same call with the cut switched off:
    the same bytes; then an invalid argument inside the tail comes back as the call's error"""
    import torch
    if expected_family(actx, 2 * 128 * 88, listed=True) != "split":
        pytest.skip("the split rule needs both secondary engines")
    n_bits, B = 2048, 88
    n = H.fixture_key()[2]
    cases = H.build_range_case(b"routing-split", [n], n_bits, B)
    oracle.set_threads(min(16, oracle.max_threads()))
    pb_h, wt_h = H.fill_batch(cases, n_bits, True, oracle)
    dev = torch.device("cuda", 0)
    actx.set_geometry(0); actx.set_enc_form("auto"); actx.set_split(True)
    try:
        pb_d, wt_d = pb_h.to(dev), wt_h.to(dev)
        st_d = torch.full((B,), 9, dtype=torch.uint8, device=dev)
        actx.range_ni_prove(pb_d.struct(), wt_d.struct(), None, None, st_d, device=True)
        assert actx.last_split() == expected_tail(B)
        v_d = torch.full((B,), 9, dtype=torch.uint8, device=dev)
        actx.range_ni_verify(pb_d.struct(), v_d, device=True)          # (reads what the prove call wrote: ordered by the events)
        assert actx.last_split() == expected_tail(B)
        actx.synchronize()
        assert not st_d.cpu().numpy().any() and bool(v_d.cpu().numpy().all())
        got = pb_d.to(None)
        actx.set_split(False)
        pb_1 = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
        pb_1.n[:] = pb_h.n; pb_1.range[:] = pb_h.range; pb_1.ciphertext[:] = pb_h.ciphertext
        actx.range_ni_prove(pb_1.struct(), wt_h.struct(), None, None, None, device=False)
        assert actx.last_split() == 0 and family_that_ran(actx) == "lat-basen"
        for f in FIELDS:
            assert np.array_equal(getattr(got, f), getattr(pb_1, f)), f
        oracle.range_ni_prove(pb_h.struct(), wt_h.struct(), None, None, None)
        for f in FIELDS:
            assert np.array_equal(getattr(pb_h, f), getattr(pb_1, f)), f
        # tampered proofs on both sides of the cut, host arrays
        actx.set_split(True)
        pb_1.resp_r1[3, 7, 0] ^= 1; pb_1.c2[70, 5, 9] ^= 4; pb_1.resp_w1[87, 100, 1] ^= 2
        v = np.full(B, 9, np.uint8)
        actx.range_ni_verify(pb_1.struct(), v, device=False)
        assert actx.last_split() == expected_tail(B)
        expect = np.ones(B, np.uint8); expect[[3, 70, 87]] = 0
        assert np.array_equal(v, expect)
    finally:
        actx.set_split(True); actx.set_geometry(0); actx.set_enc_form("auto")


def test_one_proof_verify_carries_its_transcript_hash_inside_the_enc_launch(actx, oracle):
    """A verify call whose Enc launch gives every Enc a compute unit (k_enc_basen_r2l5: one proof at 128 rows, three at 40) or every wavefront a
    SIMD (k_enc_basen_r2l: 2 - 4 proofs) runs its transcript hashes as workgroups OF that launch (csrc/zkp_api_proofs.inc range_verify_impl, zkp_diag_last_fused_hash) instead of beside it on a second
    stream: same verdicts as the oracle and as the two-stream shape, on honest proofs, on a tampered transcript (the digest changes: every
    row's kind contradicts its challenge bit with probability 1/2) and on tampered responses."""
    n_bits, kw = 2048, 64
    if actx.latency_limbs_per_lane() != 9:
        pytest.skip("the latency engine is not loaded")
    n = H.fixture_key()[2]
    oracle.set_threads(min(16, oracle.max_threads()))
    actx.set_geometry(0)
    actx.set_enc_form("auto")
    for B, ef in ((1, 128), (3, 40), (2, 64), (2, 128), (4, 128), (5, 128), (6, 128), (8, 128), (10, 128), (60, 4), (200, 1)):      # (the last two: many short transcripts — many hash workgroups in front of few Enc workgroups)
        five = 2 * ef * B <= compute_units()                       # k_enc_basen_r2l5; beyond: one wavefront per Enc, hashes aboard up to one per SIMD
        bound = 2 * ef * B                                         # (csrc/zkp_api.hip r2l_one_per_simd: three quarters of the bound exist, + 3 %)
        one_per_simd = (3 * bound + 3) // 4 + bound // 32 + B <= 4 * compute_units()
        takes = five or one_per_simd or min(bound, (3 * bound + 3) // 4 + bound // 32) <= 8 * compute_units()      # (two wavefronts per SIMD: the hashes aboard at 16 blocks per batch)
        cases = H.build_range_case(b"fused-hash-%d-%d" % (B, ef), [n], n_bits, B, ef=ef)
        pb, wt = H.fill_batch(cases, n_bits, True, oracle)
        oracle.range_generate_encrypted_pairs(pb.struct(), wt.struct())
        e = np.zeros((B, 32), np.uint8); elen = np.zeros(B, np.uint8)
        for b in range(B):
            d = oracle.fs_challenge(n_bits, ef, pb.n[0], pb.c1[b], pb.c2[b])
            e[b, :len(d)] = np.frombuffer(d, np.uint8); elen[b] = len(d)
        st = np.full(B, 9, np.uint8)
        oracle.range_generate_proof(pb.struct(), wt.struct(), e, elen, st)
        assert not st.any()
        variants = [("honest", lambda q: None),
                    ("transcript", lambda q: q.c2.__setitem__((B - 1, ef // 2, 5), q.c2[B - 1, ef // 2, 5] ^ 4)),
                    ("response", lambda q: q.resp_r1.__setitem__((0, ef - 1, 0), q.resp_r1[0, ef - 1, 0] ^ 1))]
        for name, tamper in variants:
            q = zkp.RangeBatch(n_bits, B, ef, shared_key=True)
            for f in ("n", "range", "ciphertext") + FIELDS:
                getattr(q, f)[:] = getattr(pb, f)
            tamper(q)
            want = np.full(B, 9, np.uint8)
            oracle.range_ni_verify(q.struct(), want)
            if name == "honest":
                assert (want == 1).all()
            else:
                assert (want == 0).any()
            for fused in (True, False, True):
                actx.set_fuse_hash(fused)
                v = np.full(B, 9, np.uint8)
                actx.range_ni_verify(q.struct(), v, device=False)
                assert actx.r2l_last() and actx.r2l_lanes_last() == (36 if five else 12), (B, ef, name)
                assert actx.last_fused_hash() == (fused and takes), (B, ef, name, fused)
                assert np.array_equal(v, want), (B, ef, name, fused, v, want)
    actx.set_fuse_hash(True)
