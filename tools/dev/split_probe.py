"""Would a call of 65 ... 96 proofs be faster as TWO concurrent calls on two engines (64 proofs on the mid engine, the rest on the latency
engine) than as one call on one engine?  Two contexts with their own streams, one host thread each (ctypes releases the GIL), device-resident
inputs.  python tools/dev/split_probe.py [total ...]"""
import importlib
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zkp = importlib.import_module("zk-paillier_amd")
synth = importlib.import_module("zk-paillier_amd.synth")
dev = torch.device("cuda", 0)
c1, c2, c0 = zkp.Context(0), zkp.Context(0), zkp.Context(0)


def prep(ctx, B, seed):
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=seed, device=dev)
    ctx.set_geometry(0); ctx.set_enc_form("auto")
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True); ctx.synchronize()
    return pb, wt, torch.zeros(B, dtype=torch.uint8, device=dev)


def timed(fns, reps=4):
    best = 1e9
    for _ in range(reps + 1):
        th = [threading.Thread(target=f) for f in fns[1:]]
        t0 = time.perf_counter()
        for t in th: t.start()
        fns[0]()
        for t in th: t.join()
        best = min(best, time.perf_counter() - t0)
    return round(1e3 * best, 2)


for total in [int(v) for v in sys.argv[1:]] or [72, 80, 96]:
    rec = {"total": total}
    pbA, wtA, vA = prep(c0, total, 3)
    for split_a, ga, gb in ((64, 18, 9), (48, 18, 9), (64, 18, 18), (total // 2, 9, 9)):
        a, b = split_a, total - split_a
        if b <= 0: continue
        pb1, wt1, v1 = prep(c1, a, 5)
        pb2, wt2, v2 = prep(c2, b, 7)
        c1.set_geometry(ga); c1.set_enc_form("auto" if ga == 9 else "basen")
        c2.set_geometry(gb); c2.set_enc_form("auto" if gb == 9 else "basen")
        def va(): c1.range_ni_verify(pb1.struct(), v1, device=True); c1.synchronize()
        def vb(): c2.range_ni_verify(pb2.struct(), v2, device=True); c2.synchronize()
        def pa(): c1.range_ni_prove(pb1.struct(), wt1.struct(), None, None, None, device=True); c1.synchronize()
        def pb_(): c2.range_ni_prove(pb2.struct(), wt2.struct(), None, None, None, device=True); c2.synchronize()
        rec[f"{a}@w{ga}+{b}@w{gb}"] = {"verify_concurrent": timed([va, vb]), "verify_alone": [timed([va]), timed([vb])],
                                        "prove_concurrent": timed([pa, pb_]), "prove_alone": [timed([pa]), timed([pb_])]}
        assert bool(v1.all()) and bool(v2.all())
    c0.set_geometry(0); c0.set_enc_form("auto")
    def v0(): c0.range_ni_verify(pbA.struct(), vA, device=True); c0.synchronize()
    def p0(): c0.range_ni_prove(pbA.struct(), wtA.struct(), None, None, None, device=True); c0.synchronize()
    rec["one_call_library_choice"] = {"verify": timed([v0]), "prove": timed([p0])}
    print(json.dumps(rec), flush=True)
