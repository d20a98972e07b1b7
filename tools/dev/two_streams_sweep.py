"""RangeProofNi verify (device-resident, n = 2048, one key) at B proofs with the transcript hash on the ctx's stream (ZKP_TWO_STREAMS=0) and on a
second stream beside the Enc launch (=1; the library's own rule when the variable is unset): python tools/dev/two_streams_sweep.py [sizes ...]"""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
zkp = importlib.import_module("zk-paillier_amd")
synth = importlib.import_module("zk-paillier_amd.synth")
sizes = [int(v) for v in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096]
dev = torch.device("cuda", 0)
ctx = zkp.Context(0)
if os.environ.get("ZKP_SWEEP_GEOMETRY"):
    ctx.set_geometry(int(os.environ["ZKP_SWEEP_GEOMETRY"]))      # pin an engine (36: the throughput engine, whatever the size)
for B in sizes:
    pb, wt = synth.synth_range_inputs(synth.BENCH_N, 2048, B, seed=7, device=dev)
    ctx.paillier_enc(2048, B, pb.n, 0, wt.x, wt.r, pb.ciphertext); ctx.synchronize()
    ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True); ctx.synchronize()
    v = torch.zeros(B, dtype=torch.uint8, device=dev)
    rec = {"B": B, "two_streams": os.environ.get("ZKP_TWO_STREAMS", "library rule")}
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); ctx.range_ni_verify(pb.struct(), v, device=True); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    rec["verify_ms"] = [round(x, 2) for x in ts[1:]]
    assert bool(v.all())
    print(json.dumps(rec), flush=True)
