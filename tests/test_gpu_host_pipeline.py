"""Host-pointer calls of large batches run as a pipeline of proof blocks — block k + 1 copied in and block k - 1 copied out on a second
stream under the kernels of block k (csrc/zkp_api.hip: Piped, host_blocks; opt-in through $ZKP_HOST_CHUNKS at ctx create, for hosts whose
copies are slow).  The bytes that come back must be those of the plain one-block call and of the device-pointer call, whatever the cut;
samples go to the oracle."""
import importlib
import os

import numpy as np
import pytest

import helpers as H

zkp = H.zkp
pytestmark = pytest.mark.gpu
synth = importlib.import_module("zk-paillier_amd.synth")

FIELDS = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")


def ctx_with_chunks(value):
    saved = os.environ.get("ZKP_HOST_CHUNKS")
    if value is None:
        os.environ.pop("ZKP_HOST_CHUNKS", None)
    else:
        os.environ["ZKP_HOST_CHUNKS"] = str(value)
    try:
        return zkp.Context(0)                   # (the environment is read when the ctx is created, and only then)
    finally:
        if saved is None:
            os.environ.pop("ZKP_HOST_CHUNKS", None)
        else:
            os.environ["ZKP_HOST_CHUNKS"] = saved


def host_batch_like(src, n_bits, B, shared):
    pb = zkp.RangeBatch(n_bits, B, 128, shared_key=shared)
    pb.n[:] = src.n; pb.range[:] = src.range; pb.ciphertext[:] = src.ciphertext
    return pb


@pytest.mark.parametrize("shared", [True, False], ids=["one-key", "per-proof-keys"])
def test_piped_host_calls_equal_the_plain_call(oracle, shared):
    torch = pytest.importorskip("torch")
    n_bits, B = 1024, 2208                      # rows = B * 128 above the cut-off (2048 proofs' worth); n = 1024 keeps the Enc work small
    keys = [H.test_key(1024, tag=t)[2] for t in range(7)]
    nkey = keys[0] if shared else [keys[b % 7] for b in range(B)]
    dev = torch.device("cuda", 0)
    pb_d, wt_d = synth.synth_range_inputs(nkey, n_bits, B, seed=5150 + shared, device=dev)
    plain = ctx_with_chunks(None)               # the default: one block
    piped = ctx_with_chunks(0)                  # uneven blocks: [1/4, 3/4] for verify, [1/4, 1/2, 1/4] for prove
    thirds = ctx_with_chunks(3)
    try:
        for c in (plain, piped, thirds):
            c.set_geometry(zkp.load().zkp_build_limbs_per_lane())
        plain.paillier_enc(n_bits, B, pb_d.n, 0 if shared else n_bits // 32, wt_d.x, wt_d.r, pb_d.ciphertext)
        plain.synchronize()
        src, wt = pb_d.to(None), wt_d.to(None)
        outs = []
        for c in (plain, piped, thirds):
            pb = host_batch_like(src, n_bits, B, shared)
            e = np.zeros((B, 32), np.uint8); el = np.zeros(B, np.uint8); status = np.full(B, 9, np.uint8)
            c.range_ni_prove(pb.struct(), wt.struct(), e, el, status, device=False)
            assert not status.any()
            assert c.last_host_blocks() == {id(plain): 1, id(piped): 3, id(thirds): 3}[id(c)]
            outs.append((pb, e, el))
        # the device-pointer call as the third witness
        status_d = torch.full((B,), 9, dtype=torch.uint8, device=dev)
        plain.range_ni_prove(pb_d.struct(), wt_d.struct(), None, None, status_d, device=True)
        plain.synchronize()
        ref = pb_d.to(None)
        for pb, e, el in outs:
            for f in FIELDS:
                assert np.array_equal(getattr(pb, f), getattr(ref, f)), f
            assert np.array_equal(e, outs[0][1]) and np.array_equal(el, outs[0][2]) and el.min() >= 30
        # a sample of the transcripts against the oracle
        idx = [0, 551, 552, 1655, 1656, B - 1]                   # both sides of the block boundaries of the default cut
        so = zkp.RangeBatch(n_bits, len(idx), 128, shared_key=shared)
        sw = zkp.make_range_witness(n_bits, len(idx))
        for k, b in enumerate(idx):
            so.n[0 if shared else k] = src.n[0 if shared else b]
            so.range[k] = src.range[b]; so.ciphertext[k] = src.ciphertext[b]
            for f in ("x", "r", "w1", "w2", "r1", "r2"):
                getattr(sw, f)[k] = getattr(wt, f)[b]
        oracle.set_threads(min(16, oracle.max_threads()))
        oracle.range_ni_prove(so.struct(), sw.struct(), None, None, None)
        for k, b in enumerate(idx):
            for f in FIELDS:
                assert np.array_equal(getattr(so, f)[k], getattr(ref, f)[b]), (b, f)
        # verify: tampered proofs on both sides of every boundary
        pbv = outs[1][0]
        bad = [0, 551, 552, 735, 736, 1471, 1472, 1655, 1656, B - 1]
        for k, b in enumerate(bad):
            pbv.resp_r1[b, (7 * k) % 128, 0] ^= 1
        want = np.ones(B, np.uint8); want[bad] = 0
        for c in (plain, piped, thirds):
            v = np.full(B, 9, np.uint8)
            c.range_ni_verify(pbv.struct(), v, device=False)
            assert np.array_equal(v, want)
            assert c.last_host_blocks() == {id(plain): 1, id(piped): 2, id(thirds): 3}[id(c)]
        vidx = [0, 1, 551, 552, 553, B - 1]
        sv = zkp.RangeBatch(n_bits, len(vidx), 128, shared_key=shared)
        for k, b in enumerate(vidx):
            sv.n[0 if shared else k] = pbv.n[0 if shared else b]
            for f in ("range", "ciphertext") + FIELDS:
                getattr(sv, f)[k] = getattr(pbv, f)[b]
        vo = np.full(len(vidx), 9, np.uint8)
        oracle.range_ni_verify(sv.struct(), vo)
        assert list(vo) == [int(want[b]) for b in vidx]
    finally:
        for c in (plain, piped, thirds):
            c.close()
