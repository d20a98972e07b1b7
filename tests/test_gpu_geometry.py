"""The two kernel geometries behind one ctx (include/zkp_hip.h, zkp_ctx_set_geometry): small calls go to the latency engine
(libzkp_hip_lat.so, 9 limbs per lane), large ones stay on the throughput engine (36), and both produce the same bytes — the
oracle's.  The reference's own bench shape is the small one: ONE RangeProofNi proved and verified (benches/all.rs:55-71)."""
import numpy as np
import pytest

import helpers as H
from helpers import zkp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    c = zkp.Context(0)            # automatic geometry
    yield c
    c.close()


def _rand_mod_batch(rng, count, words):
    mod = rng.integers(0, 2**32, (count, words), dtype=np.uint32); mod[:, 0] |= 1; mod[:, -1] |= 0x80000000
    base = rng.integers(0, 2**32, (count, words), dtype=np.uint32); base[:, -1] &= 0x3FFFFFFF
    exp = rng.integers(0, 2**32, (count, 2), dtype=np.uint32)
    return base, exp, mod


def test_latency_engine_is_loaded(actx):
    assert zkp.load().zkp_build_limbs_per_lane() == 36
    assert actx.latency_limbs_per_lane() == 9, "libzkp_hip_lat.so was not found next to libzkp_hip.so (run __graft_entry__.build())"


def test_automatic_choice_follows_the_size_of_the_call(actx):
    rng = np.random.default_rng(5)
    for count, want in ((8, 9), (1024, 9), (60000, 36)):      # 2048-bit moduli: the latency engine takes up to 49152 chains on 256 CUs
        base, exp, mod = _rand_mod_batch(rng, count, 64)
        out = np.zeros_like(base)
        actx.modexp(2048, 64, count, base, exp, 2, mod, 64, out)
        assert actx.last_geometry() == want, (count, actx.last_geometry())
        i = count - 1
        assert H.L.limbs_to_int(out[i]) == pow(H.L.limbs_to_int(base[i]), H.L.limbs_to_int(exp[i]), H.L.limbs_to_int(mod[i]))


def test_both_engines_and_the_oracle_agree_on_range_proof_ni(actx, oracle):
    n_bits = 2048
    n = H.fixture_key()[2]
    cases = H.build_range_case(b"geometry", [n], n_bits, 3)
    cases[2] = H.build_range_case(b"geometry-bad", [n], n_bits, 1, honest=False)[0]
    pb_o, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
    vo = np.full(3, 9, np.uint8)
    oracle.range_ni_verify(pb_o.struct(), vo)
    fields = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")
    try:
        for geometry, ran_on in ((0, 9), (36, 36), (9, 9)):
            actx.set_geometry(geometry)
            pb = zkp.RangeBatch(n_bits, 3, 128, shared_key=True)
            pb.n[:] = pb_o.n; pb.range[:] = pb_o.range; pb.ciphertext[:] = pb_o.ciphertext
            actx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=False)
            assert actx.last_geometry() == ran_on
            for f in fields:
                assert np.array_equal(getattr(pb, f), getattr(pb_o, f)), (geometry, f)
            v = np.full(3, 9, np.uint8)
            actx.range_ni_verify(pb.struct(), v, device=False)
            assert actx.last_geometry() == ran_on
            assert list(v) == list(vo) == [1, 1, 0]
    finally:
        actx.set_geometry(0)


def test_unknown_geometry_is_refused(actx):
    with pytest.raises(zkp.ZkpError, match="no engine with 12 limbs"):
        actx.set_geometry(12)
    actx.set_geometry(36); actx.set_geometry(18); actx.set_geometry(9); actx.set_geometry(0)


def test_errors_of_a_routed_call_carry_their_text(actx):
    rng = np.random.default_rng(6)
    base, exp, mod = _rand_mod_batch(rng, 4, 64)
    mod[2, 0] &= ~np.uint32(1)                                  # an even modulus
    out = np.zeros_like(base)
    with pytest.raises(zkp.ZkpError, match="even or trivial modulus"):
        actx.modexp(2048, 64, 4, base, exp, 2, mod, 64, out)
    assert actx.last_geometry() == 9
    with pytest.raises(zkp.ZkpError, match="invalid argument"):
        actx.modexp(2048, 33, 4, base, exp, 2, mod, 64, out)


def test_timing_covers_both_engines(actx):
    rng = np.random.default_rng(7)
    actx.timing_reset(True)
    try:
        for count in (16, 60000):
            base, exp, mod = _rand_mod_batch(rng, count, 64)
            out = np.zeros_like(base)
            actx.modexp(2048, 64, count, base, exp, 2, mod, 64, out)
        ms, launches, modexps = actx.timing_get()
    finally:
        actx.timing_reset(False)
    assert launches == 2 and modexps == 16 + 60000 and ms > 0


def test_ctx_on_a_caller_owned_stream():
    """zkp_ctx_create_on_stream: launches are ordered on the caller's stream (here a torch side stream), for both engines"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    c = zkp.Context(0, stream=side.cuda_stream)
    try:
        assert c.stream() == side.cuda_stream
        n = H.fixture_key()[2]
        kw = 64
        nl = torch.from_numpy(H.L.int_to_limbs(n, kw).astype(np.int32)).to(dev)
        g = torch.Generator(device=dev); g.manual_seed(3)
        for count, want in ((4, 9), (60000, 36)):      # Enc under one 2048-bit key: the secondary engines take up to 49152 items (route_latency)
            m = torch.randint(-2**31, 2**31 - 1, (count, kw), dtype=torch.int32, device=dev, generator=g); m[:, -1] &= 0x3FFFFFFF
            r = torch.randint(-2**31, 2**31 - 1, (count, kw), dtype=torch.int32, device=dev, generator=g); r[:, -1] &= 0x3FFFFFFF
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                out = torch.zeros((count, 2 * kw), dtype=torch.int32, device=dev)
                c.paillier_enc(2048, count, nl, 0, m, r, out)
                head = out[:2].clone()                       # ordered after the launch on the same stream
            side.synchronize()
            assert c.last_geometry() == want
            for i in range(2):
                mi = H.L.limbs_to_int(m[i].cpu().numpy().view(np.uint32)); ri = H.L.limbs_to_int(r[i].cpu().numpy().view(np.uint32))
                assert H.L.limbs_to_int(head[i].cpu().numpy().view(np.uint32)) == (1 + mi * n) * pow(ri, n, n * n) % (n * n)
    finally:
        c.close()


def test_ctx_lifecycle_with_both_engines_and_the_second_stream(oracle):
    """contexts come and go (each owns a twin ctx of the latency engine, lazily a second stream and its events): every one of
    them must give the same verdicts, and tearing them down must not disturb the ones still alive"""
    n_bits = 1024
    n = H.test_key(1024)[2]
    cases = H.build_range_case(b"lifecycle", [n], n_bits, 2)
    cases[1] = H.build_range_case(b"lifecycle-bad", [n], n_bits, 1, honest=False)[0]
    pb, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None)
    alive = []
    for round_ in range(12):
        c = zkp.Context(0)
        c.set_geometry((0, 36, 9)[round_ % 3])
        v = np.full(2, 9, np.uint8)
        c.range_ni_verify(pb.struct(), v, device=False)            # two-stream sequence on whichever engine serves it
        assert list(v) == [1, 0], (round_, list(v))
        alive.append(c)
        if round_ % 2:
            alive.pop(0).close()                                   # destroy an older ctx while newer ones are in use
    for c in alive:
        v = np.full(2, 9, np.uint8)
        c.range_ni_verify(pb.struct(), v, device=False)
        assert list(v) == [1, 0]
        c.close()
