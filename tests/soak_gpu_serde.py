#!/usr/bin/env python3
"""Throughput of the wire-format ingestion (by hand on a GPU box: python tests/soak_gpu_serde.py [B]): B RangeProofNi proofs
(n = 2048, 128 rows) as serde_json text -> device-resident SoA batch, then verified.  Also raw k_dec2bin / k_bin2dec rates."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from helpers import pm, L
import oracle_lib
import test_wire_format as T


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    import torch
    zkp = H.zkp
    ctx = zkp.Context(0)
    oracle = oracle_lib.Oracle()
    oracle.set_threads(min(16, oracle.max_threads()))
    n_bits, ef, kw = 2048, 128, 64
    n = H.fixture_key()[2]
    base = 8
    cases = H.build_range_case(b"serde-soak", [n], n_bits, base)
    po, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle.range_ni_prove(po.struct(), wt.struct(), None, None, None)
    pair_docs = [T.pairs_json([L.limbs_to_int(x) for x in po.c1[b]], [L.limbs_to_int(x) for x in po.c2[b]]) for b in range(base)]
    proof_docs = [T.proof_json(H.responses_from_batch(po, b)) for b in range(base)]
    pairs = [pair_docs[b % base] for b in range(B)]
    proofs = [proof_docs[b % base] for b in range(B)]
    pg = zkp.RangeBatch(n_bits, B, ef, shared_key=True, device="cuda")
    pg.n.copy_(torch.from_numpy(po.n.view(np.int32)))
    idx = [b % base for b in range(B)]
    pg.range.copy_(torch.from_numpy(po.range[idx].view(np.int32))); pg.ciphertext.copy_(torch.from_numpy(po.ciphertext[idx].view(np.int32)))
    st = torch.zeros(B, dtype=torch.uint8, device="cuda")
    for name, docs, fn in (("EncryptedPairs", pairs, ctx.lib.zkp_json_encrypted_pairs_batch), ("Proof", proofs, ctx.lib.zkp_json_range_proof_batch)):
        buf, off, ln = ctx._json_docs(docs)
        nbytes = int(ln.sum())
        s = pg.struct()
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            rc = fn(ctx.h, C.cast(buf, C.c_void_p), zkp.capi.ptr(off), zkp.capi.ptr(ln), C.byref(s), zkp.capi.ptr(st), zkp.ZKP_F_DEVICE_PTRS)
            ctx.synchronize(); dt = time.time() - t0
        assert rc == 0 and int(st.sum()) == 0
        print(f"{name}: {B} documents, {nbytes / 1e6:.1f} MB of JSON -> SoA in {dt * 1e3:.1f} ms = {nbytes / dt / 1e9:.2f} GB/s, {B / dt:.0f} proofs/s", flush=True)
    v = torch.zeros(B, dtype=torch.uint8, device="cuda")
    ctx.range_ni_verify(pg.struct(), v, device=True); ctx.synchronize()
    print("verified after ingestion: all accepted =", bool((v == 1).all()))
    # raw conversion kernels, host buffers (PCIe included)
    d = pm.Drbg(b"serde-raw")
    vals = [d.below(1 << 4096) for _ in range(4096)]
    src = np.tile(L.ints_to_limbs(vals, 128), (16, 1))
    t0 = time.time(); out = ctx.limbs_to_decimal(src); dt = time.time() - t0
    assert out[:4096] == [str(x).encode() for x in vals]
    print(f"limbs -> decimal, {len(src)} x 4096-bit incl. PCIe and Python slicing: {dt * 1e3:.0f} ms")


if __name__ == "__main__":
    main()
