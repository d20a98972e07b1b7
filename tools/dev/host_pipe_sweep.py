"""RangeProofNi prove / verify at B proofs on HOST arrays (what a Rust caller hands over) under several cuts of the call into proof
blocks ($ZKP_HOST_CHUNKS at ctx create: 1 = the plain path, unset = the library's rule), beside the device-resident call.
python tools/dev/host_pipe_sweep.py [B]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zkp = importlib.import_module("zk-paillier_amd")
synth = importlib.import_module("zk-paillier_amd.synth")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_bits = 2048
dev = torch.device("cuda", 0)


def ctx_with(chunks):
    if chunks is None:
        os.environ.pop("ZKP_HOST_CHUNKS", None)
    else:
        os.environ["ZKP_HOST_CHUNKS"] = str(chunks)
    c = zkp.Context(0)
    os.environ.pop("ZKP_HOST_CHUNKS", None)
    return c


base = ctx_with(None)
pb_d, wt_d = synth.synth_range_inputs(synth.BENCH_N, n_bits, B, seed=4, device=dev)
base.paillier_enc(n_bits, B, pb_d.n, 0, wt_d.x, wt_d.r, pb_d.ciphertext)
status = torch.zeros(B, dtype=torch.uint8, device=dev)
verdict = torch.zeros(B, dtype=torch.uint8, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    base.range_ni_prove(pb_d.struct(), wt_d.struct(), None, None, status, device=True); base.synchronize()
    t1 = time.perf_counter()
    base.range_ni_verify(pb_d.struct(), verdict, device=True); base.synchronize()
    t2 = time.perf_counter()
print(json.dumps({"B": B, "device_resident": True, "prove_ms": round(1e3 * (t1 - t0), 1), "verify_ms": round(1e3 * (t2 - t1), 1)}), flush=True)
ref, wt = pb_d.to(None), wt_d.to(None)
base.close()
for chunks in (None, 0, 2, 3, 4):
    c = ctx_with(chunks)
    pb = zkp.RangeBatch(n_bits, B, 128, shared_key=True)
    pb.n[:] = ref.n; pb.range[:] = ref.range; pb.ciphertext[:] = ref.ciphertext
    st = np.zeros(B, np.uint8); v = np.zeros(B, np.uint8)
    rec = {"B": B, "ZKP_HOST_CHUNKS": "unset (one block)" if chunks is None else chunks}
    for rep in range(2):                     # (the first pass warms the staging blocks)
        t0 = time.perf_counter()
        c.range_ni_prove(pb.struct(), wt.struct(), None, None, st, device=False)
        t1 = time.perf_counter()
        nb_p = c.last_host_blocks()
        c.range_ni_verify(pb.struct(), v, device=False)
        t2 = time.perf_counter()
        rec.update(prove_ms=round(1e3 * (t1 - t0), 1), verify_ms=round(1e3 * (t2 - t1), 1), prove_blocks=nb_p, verify_blocks=c.last_host_blocks())
    rec["equal_to_device_resident"] = bool(all(np.array_equal(getattr(pb, f), getattr(ref, f)) for f in ("c1", "c2", "resp_w1", "resp_r1")) and v.all())
    print(json.dumps(rec), flush=True)
    c.close()
