"""GPU parity: CompositeDLogProof prove / verify (wi_dlog_proof.rs:46-91) against the oracle and the goldens."""
import json
import os

import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

pytestmark = pytest.mark.gpu


def test_dlog_goldens(ctx):
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dlog.json")))
    iv = lambda s: int(s, 16)
    B = len(gold)
    arr = lambda k, w: L.ints_to_limbs([iv(g[k]) for g in gold], w)
    N, g, ni, s, r = arr("N", 64), arr("g", 64), arr("ni", 64), arr("secret", 8), arr("r", 16)
    x = np.zeros((B, 64), np.uint32); y = np.zeros((B, 24), np.uint32)
    ctx.dlog_prove(2048, 768, B, N, g, ni, s, r, x, y)
    assert L.limbs_to_ints(x) == [iv(k["x"]) for k in gold] and L.limbs_to_ints(y) == [iv(k["y"]) for k in gold]
    v = np.full(B, 9, np.uint8)
    ctx.dlog_verify(2048, 768, B, N, g, ni, x, y, v)
    assert list(v) == [k["verdict"] for k in gold]


@pytest.mark.parametrize("n_bits", [1024, 2048])
def test_dlog_matches_oracle(ctx, oracle, n_bits):
    kw, yb = n_bits // 32, 768
    d = pm.Drbg(b"gpu-dlog-%d" % n_bits)
    rows = []
    for t in range(3):
        p, q, N = H.test_key(n_bits, tag=20 + t)
        g = d.range(2, N - 1); s = d.bits(256)
        rows.append((N, g, pow(pow(g, -1, N), s, N), s, d.bits(512)))     # honest (wi_dlog_proof.rs:117-141)
        rows.append((N, g, pow(g, s, N), s, d.bits(512)))                 # +s instead of -s (:145-168)
    p, q, N = H.test_key(n_bits, tag=20)
    rows.append((N, p, rows[0][2], rows[0][3], d.bits(512)))              # gcd(g, N) != 1 -> panic in the reference (:72)
    rows.append((N, rows[0][1], q * 3, rows[0][3], d.bits(512)))          # gcd(ni, N) != 1 (:73)
    rows.append(((1 << 128) - 159, 5, 7, 3, d.bits(512)))                 # N <= 2^128 (:69)
    rows.append((N, 0, rows[0][2], rows[0][3], d.bits(512)))              # g = 0: gcd(0, N) = N
    # g == N modulo 2^64 (and a multiple of p next to it): the first difference of the binary GCD has zero low words
    g_z = N - (d.bits(300) << 64); s_z = d.bits(256)
    rows.append((N, g_z, pow(pow(g_z, -1, N), s_z, N), s_z, d.bits(512)))
    rows.append((N, N - (p << 96), rows[0][2], rows[0][3], d.bits(512)))
    B = len(rows)
    N_, g_, ni_ = (L.ints_to_limbs([r[i] for r in rows], kw) for i in range(3))
    s_ = L.ints_to_limbs([r[3] for r in rows], 8); r_ = L.ints_to_limbs([r[4] for r in rows], 16)
    xo, yo = oracle.dlog_prove(n_bits, yb, N_, g_, ni_, s_, r_)
    xg = np.zeros_like(xo); yg = np.zeros_like(yo)
    ctx.dlog_prove(n_bits, yb, B, N_, g_, ni_, s_, r_, xg, yg)
    assert np.array_equal(xo, xg) and np.array_equal(yo, yg)
    # extra verify-side cases built from the honest proof 0: x + N (non-canonical x never equals a residue), y + 1
    N2, g2, ni2, x2, y2 = (np.concatenate([a, a[:1], a[:1]]) for a in (N_, g_, ni_, xo, yo))
    v = L.limbs_to_int(x2[B]) + rows[0][0]
    if v.bit_length() <= n_bits:
        x2[B] = L.int_to_limbs(v, kw)
    y2[B + 1, 0] ^= 1
    vo = oracle.dlog_verify(n_bits, yb, N2, g2, ni2, x2, y2)
    vg = np.full(B + 2, 9, np.uint8)
    ctx.dlog_verify(n_bits, yb, B + 2, N2, g2, ni2, x2, y2, vg)
    assert np.array_equal(vo, vg), (list(vo), list(vg))
    assert vo[0] == zkp.VERDICT_ACCEPT and vo[1] == zkp.VERDICT_REJECT and vo[6] == zkp.VERDICT_MALFORMED and vo[8] == zkp.VERDICT_MALFORMED
    # the same proofs tiled to a batch that fills the GPU: one launch per exponentiation, pre-checks on the one stream (the
    # small call above took the merged launch with the pre-checks on the second stream, zkp_api_proofs.inc:dlog_verify_impl)
    tiles = 3000
    big = [np.tile(a, (tiles, 1)) for a in (N2, g2, ni2, x2, y2)]
    vb = np.full((B + 2) * tiles, 9, np.uint8)
    ctx.dlog_verify(n_bits, yb, (B + 2) * tiles, *big, vb)
    assert np.array_equal(vb, np.tile(vo, tiles))
