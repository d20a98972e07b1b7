import os, sys, time, importlib
import numpy as np, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import helpers as H
zkp = H.zkp
synth = importlib.import_module("zk-paillier_amd.synth")
ctx = zkp.Context(0); dev = torch.device("cuda", 0)
pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, 1, seed=7, device=dev)
ctx.paillier_enc(2048, 1, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
pbh = pb.to(None); wth = wt.to(None); v = np.zeros(1, np.uint8)
for rep in range(5):
    ctx.range_ni_prove(pbh.struct(), wth.struct(), None, None, None, device=False)
    ctx.range_ni_verify(pbh.struct(), v, device=False)
print(v)
