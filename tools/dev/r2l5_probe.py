"""One to eight RangeProofNi proofs (device-resident, n = 2048, the fixture key) on the latency engine's right-to-left ladder under its two
lane geometries — one wavefront of five 12-lane groups per Enc, five wavefronts of 36 lanes per Enc — and under the library's own choice:
prove / verify ms (best of 5), the transcripts of the two compared byte for byte.  python tools/dev/r2l5_probe.py [sizes ...]"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zkp = importlib.import_module("zk-paillier_amd")
synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0)
dev = torch.device("cuda", 0)
sizes = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 6, 8]
FIELDS = ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")


def best_of(fn, reps=5):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(1e3 * best, 2)


for B in sizes:
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
    ctx.set_geometry(0); ctx.set_enc_form("auto"); ctx.set_r2l(1); ctx.set_r2l_lanes(0)
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"B": B}
    snaps = {}
    for name, lanes, forced in (("one_wavefront_12x6", 12, True), ("five_wavefronts_36x2", 36, True), ("auto", 0, False)):
        ctx.set_geometry(9 if forced else 0); ctx.set_r2l(2 if forced else 1); ctx.set_r2l_lanes(lanes)
        rec[name] = [best_of(lambda: ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True)),
                     best_of(lambda: ctx.range_ni_verify(pb.struct(), v, device=True))]
        rec[name + "_ran"] = f"w{ctx.last_geometry()} r2l={int(ctx.r2l_last())} lanes={ctx.r2l_lanes_last()}"
        assert bool(v.all()), (B, name)
        snaps[name] = [getattr(pb, f).clone() for f in FIELDS]
    rec["transcripts_equal"] = all(torch.equal(x, y) for x, y in zip(snaps["one_wavefront_12x6"], snaps["five_wavefronts_36x2"])) and \
        all(torch.equal(x, y) for x, y in zip(snaps["one_wavefront_12x6"], snaps["auto"]))
    print(json.dumps(rec), flush=True)
