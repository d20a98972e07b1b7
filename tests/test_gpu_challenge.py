"""utils::compute_digest + the Fiat-Shamir glue (src/zkproofs/utils.rs:9-22, range_proof_ni.rs:58-61) through its own entry point,
against the oracle: both device hash kernels (one lane per proof for large calls, one wavefront per proof for small ones) on
transcripts whose values have every byte length — zero, a few bytes, leading zero bytes and words, full width."""
import json
import os

import numpy as np
import pytest

import helpers as H
from helpers import L, pm, zkp

pytestmark = pytest.mark.gpu


def _ragged_values(rng, count, words):
    """values whose minimal big-endian encodings have all kinds of lengths (N1: zero is the single byte 00)"""
    out = []
    for i in range(count):
        kind = i % 8
        if kind == 0: v = 0
        elif kind == 1: v = int(rng.integers(1, 256))                                   # one byte
        elif kind == 2: v = int(rng.integers(1, 2**31)) << (8 * int(rng.integers(0, 4 * words - 4)))   # trailing zero bytes, odd lengths
        elif kind == 3: v = (1 << (32 * words)) - 1                                     # full width, all ones
        elif kind == 4: v = int.from_bytes(rng.bytes(4 * words - int(rng.integers(1, 9))), "big")     # 1..8 leading zero bytes
        elif kind == 5: v = int.from_bytes(rng.bytes(int(rng.integers(1, 4 * words))), "big")          # any length
        elif kind == 6: v = 1 << (32 * int(rng.integers(1, words)))                     # a power of 2^32: one byte and zero words below
        else: v = int.from_bytes(rng.bytes(4 * words), "big")
        out.append(v)
    return out


@pytest.mark.parametrize("n_bits,ef,batch", [(1024, 128, 5), (2048, 128, 3), (1024, 7, 70), (1024, 0, 2), (2048, 33, 1100)])
def test_challenge_matches_the_oracle(ctx, oracle, n_bits, ef, batch):
    kw = n_bits // 32
    rng = np.random.default_rng(n_bits + ef + batch)
    pb = zkp.RangeBatch(n_bits, batch, ef, shared_key=False)
    ns = [v | 1 for v in _ragged_values(rng, batch, kw)]
    ns[0] = H.test_key(1024)[2] if n_bits == 1024 else H.fixture_key()[2]
    want = []
    for b in range(batch):
        c1 = _ragged_values(rng, ef, 2 * kw)
        c2 = list(reversed(_ragged_values(rng, ef, 2 * kw)))
        pb.n[b] = L.int_to_limbs(ns[b], kw)
        if ef:
            pb.c1[b] = L.ints_to_limbs(c1, 2 * kw); pb.c2[b] = L.ints_to_limbs(c2, 2 * kw)
        if b < 6 or b % 97 == 0:
            want.append((b, pm.fs_challenge(ns[b], c1, c2)))
            if ef:
                assert oracle.fs_challenge(n_bits, ef, pb.n[b], pb.c1[b], pb.c2[b]) == want[-1][1]
    e = np.zeros((batch, 32), np.uint8); elen = np.zeros(batch, np.uint8)
    ctx.range_challenge(pb.struct(), e, elen, device=False)
    for b, w in want:
        assert bytes(e[b, :elen[b]]) == w, (b, bytes(e[b, :elen[b]]).hex(), w.hex())
        assert not e[b, elen[b]:].any()


def test_challenge_golden_leading_zero(ctx):
    z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "digest_kat.json")))["leading_zero_challenge"]
    n, c1, c2 = int(z["n"], 16), [int(v, 16) for v in z["c1"]], [int(v, 16) for v in z["c2"]]
    pb = zkp.RangeBatch(1024, 1, len(c1), shared_key=True)
    pb.n[0] = L.int_to_limbs(n, 32)
    pb.c1[0] = L.ints_to_limbs(c1, 64); pb.c2[0] = L.ints_to_limbs(c2, 64)
    e = np.zeros((1, 32), np.uint8); elen = np.zeros(1, np.uint8)
    ctx.range_challenge(pb.struct(), e, elen, device=False)
    assert bytes(e[0, :elen[0]]).hex() == z["e"] and elen[0] < 32
