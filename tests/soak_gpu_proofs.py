#!/usr/bin/env python3
"""Randomised soak of RangeProofNi prove/verify against the oracle (by hand on a GPU box:
python tests/soak_gpu_proofs.py [rounds]).  Every round proves B proofs (honest and dishonest witnesses, shared and
per-proof keys), compares the full transcripts byte for byte, then applies random tampering to random fields of random
rows and compares the verdict vectors."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from helpers import pm, L
import oracle_lib

FIELDS = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2", "range", "ciphertext")


def run(ctx, oracle, seed=1000, rounds=3, deadline=None, log=print, tag=b"soakp", n_bits=1024, batches=(48, 1, 5, 300), geometries=(0, 36, 9), forms=("auto", "basen", "n2")):
    """`rounds` rounds (or until `deadline`, time.monotonic()) of prove / tamper / verify; -> proofs checked"""
    import time
    zkp = H.zkp
    keys = [H.test_key(n_bits // 2 if n_bits == 1024 else n_bits, tag=t)[2] for t in range(6)]      # (n = 2048: full-size keys, the ones the base-n kernels take)
    total = 0
    for rd in range(rounds):
        rng = np.random.default_rng(seed + rd)
        B = batches[rd % len(batches)]                     # one proof ... a batch that takes the one-stream verify sequence
        ctx.set_geometry(geometries[rd % len(geometries)])  # automatic choice / pinned to either engine
        ctx.set_enc_form(forms[(rd // len(geometries)) % len(forms)])      # the library's own choice of Enc form / either form pinned (n = 2048 and up)
        shared = bool(rd % 2)
        klist = [keys[rd % 6]] if shared else [keys[(rd + b) % 6] for b in range(B)]
        cases = H.build_range_case(tag + b"-%d-%d" % (seed, rd), klist, n_bits, B, shared=shared)
        for b in range(0, B, 7):
            cases[b] = H.build_range_case(tag + b"-bad-%d-%d-%d" % (seed, rd, b), [cases[b]["n"]], n_bits, 1, honest=False)[0]
        pb_o, wt = H.fill_batch(cases, n_bits, shared, oracle)
        pb_g = zkp.RangeBatch(n_bits, B, 128, shared_key=shared)
        pb_g.n[:] = pb_o.n; pb_g.range[:] = pb_o.range; pb_g.ciphertext[:] = pb_o.ciphertext
        oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
        ctx.range_ni_prove(pb_g.struct(), wt.struct(), None, None, None, device=False)
        for f in FIELDS[:8]:
            assert np.array_equal(getattr(pb_o, f), getattr(pb_g, f)), (seed, rd, f)
        # random tampering: ~half of the proofs get 1..3 random edits
        for b in range(B):
            if rng.random() < 0.5:
                continue
            for _ in range(int(rng.integers(1, 4))):
                f = FIELDS[int(rng.integers(0, len(FIELDS)))]
                a = getattr(pb_g, f)
                if a.ndim == 2 and a.dtype == np.uint8:          # resp_kind / resp_j
                    a[b, int(rng.integers(0, 128))] = int(rng.integers(0, 4))
                elif a.ndim == 2:                               # range / ciphertext
                    a[b, int(rng.integers(0, a.shape[1]))] ^= np.uint32(1 << int(rng.integers(0, 32)))
                else:
                    a[b, int(rng.integers(0, 128)), int(rng.integers(0, a.shape[2]))] ^= np.uint32(1 << int(rng.integers(0, 32)))
        vo = np.full(B, 9, np.uint8); vg = np.full(B, 9, np.uint8)
        oracle.range_ni_verify(pb_g.struct(), vo)
        ctx.range_ni_verify(pb_g.struct(), vg, device=False)
        assert np.array_equal(vo, vg), (seed, rd, list(vo), list(vg))
        total += B
        log("round", rd, "B", B, "ran on", ctx.last_geometry(), "limbs per lane, ok: accepted", int((vo == 1).sum()), "rejected", int((vo == 0).sum()), "of", B)
        if deadline is not None and time.monotonic() > deadline:
            break
    ctx.set_geometry(0)
    ctx.set_enc_form("auto")
    return total


def main():
    """python tests/soak_gpu_proofs.py [rounds] [n_bits]; n_bits = 2048 (every third round pins the base-n form, every third the n^2-sized kernels)
    soaks the base-n kernels, shared and per-proof keys in turn"""
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    ctx = H.zkp.Context(0)
    oracle = oracle_lib.Oracle()
    oracle.set_threads(min(16, oracle.max_threads()))
    batches = (48, 1, 5, 300) if n_bits == 1024 else (24, 3, 40, 17)
    print("PROOF SOAK OK", run(ctx, oracle, 1000, rounds, log=lambda *a: print(*a, flush=True), n_bits=n_bits, batches=batches,
                               geometries=(0, 36, 9) if n_bits == 1024 else (36,)))


if __name__ == "__main__":
    main()
