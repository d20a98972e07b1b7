#!/usr/bin/env python3
"""bench.py — RangeProofNi verify (headline) and prove throughput at n=2048, batch=4096 proofs per GPU.

A "step" of the headline metric is one pass of zkp_range_ni_verify_batch over one batch of
B synthetic proofs already resident in HBM (BASELINE.json configs[1]); the prove leg
(configs[2]) is timed the same way and reported beside it.  One process per GPU: proof indices
are sharded by rank (weak scaling: B proofs per rank), no collective on the data path, and ONE
RCCL all-gather per step (zk-paillier_amd/shard.py) reassembles the verdict vector (and the
ciphertext slabs of the prove leg) inside the timed region.  The process group is always
initialised — at N=1 the gather degenerates to a copy but the same RCCL code runs.

`python bench.py --gpus N` with N > 1 starts the N ranks itself (torch.distributed.run on
127.0.0.1); under an external launcher (RANK / WORLD_SIZE set) it checks WORLD_SIZE == N and
exits with status 2 otherwise.

Prints ONE JSON line on rank 0.  See DESIGN.md §6 for the definitions of roofline/cpu_baseline."""
import argparse
import glob
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The VALU roofline is measured, the guides list no integer peak: v_mad_u64_u32, 16 independent accumulators, 8 waves/SIMD,
# 256 CUs (csrc/microbench).  profiles/mad_sustained_r02.jsonl: kernels of 0.13 s, 1 s and 4 s all issue 3.354-3.361e13
# lane-MAC/s (4.68 cycles per wave64 instruction at the nominal 2.4 GHz): that SUSTAINED rate is the peak the launches of this
# bench (1.6-19 s each) are priced against.  Kernels of 16 ms read 3.474e13 (profiles/valu_rates_long_r01.jsonl; round 1 used
# that figure), kernels of 1 ms 3.19e13: the shader clock is not constant.  Both fractions are reported.
PEAK_LIMB_MAC_PER_S = 3.361e13
PEAK_LIMB_MAC_PER_S_16MS_KERNELS = 3.474e13
HBM_PEAK_GBS = 8000.0


def enc_limb_macs(n_bits):
    """algorithmic 32x32->64 limb-MACs of one Enc: 1.2*n_bits modmuls x (2L^2+L), L = 2*n_bits/32"""
    Lw = 2 * n_bits // 32
    return 1.2 * n_bits * (2 * Lw * Lw + Lw)


def modexp_limb_macs(mod_bits, exp_bits):
    Lw = mod_bits // 32
    return 1.2 * exp_bits * (2 * Lw * Lw + Lw)


def usable_cores(omp_max):
    """host cores this process may really use: min(affinity, cgroup cpu quota, OpenMP max)"""
    n = min(omp_max, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def pmc_traffic_per_modexp(kernel_substr):
    """HBM-side bytes per modexp of `kernel_substr` from the newest aggregated PMC file under profiles/ (separate rocprofv3 --pmc
    passes of `bench.py --pmc-shape`, profiles/collect_pmc.sh + aggregate_pmc.py).  gfx950 correction of the guide (MI355X_MICROARCH.md
    §HBM): FETCH_SIZE under-reads wide coalesced reads by 2x -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  None if no file."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), key=lambda f: (os.path.basename(f)[:3], os.path.getmtime(f)))
    for f in reversed(files):
        try:
            data = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, rec in data.items():
            d = rec.get("_derived") if isinstance(rec, dict) else None
            # (kernel names carry further template arguments in later builds: "k_enc<4, true" matches "k_enc<4, true, false>")
            if kernel_substr.rstrip(">") in k and d and "modexps_in_these_dispatches" in d and "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
                return (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0 / d["modexps_in_these_dispatches"], os.path.relpath(f, ROOT)
    return None, None


class GpuEngine:
    """the product path: the C ABI on device-resident buffers (raises if the HIP library / a gfx950 GPU is missing)"""

    def __init__(self, ctx, torch):
        self.ctx, self.torch = ctx, torch

    def prove(self, pb, wt):
        self.ctx.range_ni_prove(pb.struct(), wt.struct(), None, None, None, device=True)

    def verify(self, pb, verdict):
        self.ctx.range_ni_verify(pb.struct(), verdict, device=True)

    def before_collective(self):
        self.ctx.synchronize()             # the engine works on its own stream; the collective runs on torch's

    def after_collective(self):
        self.torch.cuda.synchronize()      # the next step overwrites c1/c2/verdict on the engine's stream


def make_steps(engine, pb, wt, verdict, world):
    """the two timed step functions.  `engine` supplies prove / verify on this rank's block of proofs; the gather of the output
    slabs goes through zk-paillier_amd/shard.py (RCCL on GPUs; tests/test_distributed_gloo.py drives these same functions over gloo)."""
    shard = importlib.import_module("zk-paillier_amd.shard")
    out = {}

    def prove_step():
        engine.prove(pb, wt)
        engine.before_collective()
        out["c1"] = shard.all_gather_slabs(pb.c1, world)
        out["c2"] = shard.all_gather_slabs(pb.c2, world)
        engine.after_collective()

    def verify_step():
        engine.verify(pb, verdict)
        engine.before_collective()
        out["verdict"] = shard.all_gather_slabs(verdict, world)
        engine.after_collective()

    return prove_step, verify_step, out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="proofs per GPU")
    ap.add_argument("--n-bits", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=64, help="proofs verified by the all-cores CPU baseline (0 = skip the CPU legs)")
    ap.add_argument("--no-prove-leg", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the legs for the other BASELINE.json configurations")
    ap.add_argument("--big-batch", type=int, default=4096, help="proofs of the n=4096 leg (configs[4]); 0 = skip")
    ap.add_argument("--distinct-batch", type=int, default=4096, help="proofs of the distinct-keys leg (SURVEY 8(d) config 3); 0 = skip")
    ap.add_argument("--no-pcie-leg", action="store_true")
    ap.add_argument("--pmc-shape", choices=["enc2048", "enc2048keys", "enc4096", "ck2048"], default=None,
                    help="run ONE short launch shape only (for rocprofv3 --pmc passes, profiles/collect_pmc.sh)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        # self-launch: one process per GPU on this node
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist
    zkp = importlib.import_module("zk-paillier_amd")
    synth = importlib.import_module("zk-paillier_amd.synth")

    if "RANK" not in os.environ:            # N = 1 started plainly: a one-rank process group in this process
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port())})
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs cuda:{local_rank} but {torch.cuda.device_count()} GPUs are visible", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)          # backend "nccl" IS RCCL on ROCm
    ctx = zkp.Context(local_rank)          # raises if the HIP library / a gfx950 GPU is missing
    lpl = int(zkp.load().zkp_build_limbs_per_lane())
    ctx.set_geometry(lpl)                  # every batch leg runs on the throughput engine, whatever --batch says (the latency engine is measured in configs[0])
    engine = GpuEngine(ctx, torch)

    B, n_bits, EF = args.batch, args.n_bits, 128
    kw = n_bits // 32
    n = synth.BENCH_N
    assert n_bits == 2048, "the bench key is the reference's 2048-bit fixture"

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    def barrier():
        sync()
        dist.barrier()
        sync()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        ctx.timing_reset(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        kms, launches, modexps = ctx.timing_get()
        ctx.timing_reset(False)
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), kms, launches, modexps

    def enc_roofline(kms, launches, modexps, nb, kernel, extra_note=""):
        ach = modexps * enc_limb_macs(nb) / (kms * 1e-3) if kms else 0.0
        per, src = pmc_traffic_per_modexp(kernel.split(" (")[0])            # the kernel's name as rocprofv3 prints it
        per_launch = modexps / max(launches, 1)
        bytes_per_enc = 4 * (nb // 32) * 4 + 8          # r, m (kw words each) + expected ciphertext (2kw) + 8 B work item
        return {"bound": "valu", "achieved": ach / 1e12, "peak": PEAK_LIMB_MAC_PER_S / 1e12, "unit": "Tlimb-MAC/s", "frac": ach / PEAK_LIMB_MAC_PER_S,
                "peak_note": "sustained v_mad_u64_u32 issue rate measured with 1-4 s kernels (profiles/mad_sustained_r02.jsonl)",
                "frac_vs_16ms_kernel_peak": ach / PEAK_LIMB_MAC_PER_S_16MS_KERNELS, "peak_16ms_kernels": PEAK_LIMB_MAC_PER_S_16MS_KERNELS / 1e12,
                "traffic": per * per_launch if per else None,
                "traffic_note": (f"bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 per Enc from {src} (separate rocprofv3 --pmc passes of this build; FETCH_SIZE doubled "
                                 f"per the guide's gfx950 note) x Enc of the launch; algorithmic operand bytes are ~{bytes_per_enc} B per Enc" if per else
                                 "no aggregated PMC file for this kernel under profiles/") + extra_note,
                "kernel": kernel, "kernel_ms_per_launch": kms / max(launches, 1), "modexps_per_launch": per_launch,
                "hbm": {"achieved": modexps * bytes_per_enc / (kms * 1e-3) / 1e9 if kms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s"}}

    # ---- single launch shapes for PMC passes (no timing legs, no CPU work)
    if args.pmc_shape:
        run_pmc_shape(args, ctx, synth, torch, dev, sync)
        dist.barrier(); dist.destroy_process_group()
        return

    # ---- inputs (untimed): witnesses in HBM, ciphertext = Enc(x, r) by the engine itself
    pb, wt = synth.synth_range_inputs(n, n_bits, B, seed=1234 + rank, device=dev)
    sync()
    ctx.paillier_enc(n_bits, B, pb.n, 0, wt.x, wt.r, pb.ciphertext)
    sync()
    verdict = torch.zeros(B, dtype=torch.uint8, device=dev)
    prove_step, verify_step, gathered = make_steps(engine, pb, wt, verdict, world)

    # ---- prove leg (also produces the proofs the verify leg consumes)
    prove = None
    if args.no_prove_leg:
        engine.prove(pb, wt); sync()
    else:
        dt, kms, launches, modexps = timed(prove_step, args.steps, args.warmup)
        prove = {"value": B * world * args.steps / dt, "unit": "proofs/s", "ms_per_step": 1e3 * dt / args.steps,
                 "enc_kernel_ms_per_launch": kms / max(launches, 1), "launches": launches,
                 "achieved_limb_mac_per_s": modexps * enc_limb_macs(n_bits) / (kms * 1e-3) if kms else None,
                 "frac": modexps * enc_limb_macs(n_bits) / (kms * 1e-3) / PEAK_LIMB_MAC_PER_S if kms else None}
    # tamper every 64th proof (one bit of resp_r1 in row 0): those must be rejected, all others accepted
    tampered = torch.arange(0, B, 64, device=dev)
    pb.resp_r1[tampered, 0, 0] ^= 1
    expect = torch.ones(B, dtype=torch.uint8, device=dev)
    expect[tampered] = 0
    sync()

    # ---- verify leg (headline)
    dt, kms, launches, modexps = timed(verify_step, args.steps, args.warmup)
    sync()
    ok = bool(torch.equal(verdict, expect))
    ok = ok and bool(torch.equal(gathered["verdict"].view(world, B)[rank], expect))
    if "c1" in gathered:
        ok = ok and bool(torch.equal(gathered["c1"].view(world, B, *pb.c1.shape[1:])[rank], pb.c1))
    value = B * world * args.steps / dt
    roofline = enc_roofline(kms, launches, modexps, n_bits, f"k_enc<{144 // lpl}, true> (fused Enc-and-compare; {144 // lpl} lanes x {lpl} limbs per 4096-bit integer; sliding-window ladder)")
    ms_per_step = 1e3 * dt / args.steps
    gathered.clear()

    cpu = pcie = other = None
    if rank == 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        if args.cpu_sample > 0:
            cpu, same = cpu_baseline(args, pb, wt, verdict, np)
            ok = ok and same
        if not args.no_pcie_leg:
            pcie, same = pcie_leg(ctx, pb, expect, np, B)
            ok = ok and same
        if not args.no_other_configs:
            other, same = other_configs(args, ctx, synth, torch, dev, sync, pb, wt, rank, enc_roofline, lpl, np)
            ok = ok and same

    if rank == 0:
        out = {"metric": "RangeProofNi proofs/sec + verifies/sec, n=2048, batch=4096 per GPU (value = verifies/sec; proofs/sec in prove.value)", "value": value, "unit": "verifies/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (29-bit limbs, u64 accumulate)",
               "data": "synthetic", "verdicts_ok": ok,
               "config": {"workload": f"BASELINE.json configs[1]: batch={B} RangeProofNi verify per GPU, n={n_bits} (reference fixture key), "
                                      f"128 rows/proof, 1/64 of the proofs tampered; prove leg = configs[2]",
                          "parallelism": f"proof-index sharding x{world}, one RCCL all-gather per step of verdicts (verify) and c1/c2 slabs (prove) via zk-paillier_amd/shard.py"},
               "prove": prove, "roofline": roofline, "cpu_baseline": cpu, "pcie_inclusive": pcie, "other_configs": other}
        # RCCL writes a version banner through C stdio when the communicator is created; push it out first so that the
        # JSON line is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)


def cpu_baseline(args, pb, wt, verdict, np):
    """The C/GMP oracle (the library the reference's BigInt bottoms out in) on BOUNDED samples of the same batch, rank 0 at N=1 only:
    verify and prove on all usable cores (OpenMP over (proof,row): the analogue of the reference's rayon par_iter), and BASELINE
    configs[0] — ONE proof proved and verified (benches/all.rs:55-71) — on one thread and on all cores."""
    import oracle_lib   # test infrastructure: used here only as the reported CPU baseline / checker
    oracle = oracle_lib.Oracle()
    B = pb.batch
    S = min(args.cpu_sample, B)
    threads = usable_cores(oracle.max_threads())
    host = pb.slice(0, S).to(None)            # contains tampered proof 0
    oracle.set_threads(threads)
    vo = np.zeros(S, np.uint8)
    t0 = time.perf_counter()
    oracle.range_ni_verify(host.struct(), vo)
    t_v = time.perf_counter() - t0
    same = bool(np.array_equal(vo, verdict[:S].cpu().numpy()))
    # prove: half the sample (a prove is 256 Enc against ~192 of a verify); outputs must equal the GPU's
    P = max(1, S // 2)
    hp = pb.slice(0, P).to(None); hw = wt.slice(0, P).to(None)
    ref_c1 = hp.c1.copy()
    hp.c1[:] = 0
    t0 = time.perf_counter()
    oracle.range_ni_prove(hp.struct(), hw.struct(), None, None, None)
    t_p = time.perf_counter() - t0
    same_p = bool(np.array_equal(hp.c1, ref_c1))
    # configs[0]: one proof, prove + verify, 1 thread and all cores
    h1 = pb.slice(1, 2).to(None); w1 = wt.slice(1, 2).to(None)
    lat = {}
    for label, th in (("1_thread", 1), ("all_cores", threads)):
        oracle.set_threads(th)
        v1 = np.zeros(1, np.uint8)
        t0 = time.perf_counter()
        oracle.range_ni_prove(h1.struct(), w1.struct(), None, None, None)
        t1 = time.perf_counter()
        oracle.range_ni_verify(h1.struct(), v1)
        t2 = time.perf_counter()
        lat[label] = {"threads": th, "prove_ms": 1e3 * (t1 - t0), "verify_ms": 1e3 * (t2 - t1), "prove_plus_verify_ms": 1e3 * (t2 - t0), "accepted": bool(v1[0] == 1)}
    one = lat["1_thread"]
    cpu = {"value": S / t_v, "unit": "verifies/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"oracle (C + GMP 6.2.1 mpz_powm, OpenMP over (proof,row)) verifying proofs 0..{S-1} of the same batch in {t_v:.2f}s; verdicts equal to GPU: {same}",
           "prove": {"value": P / t_p, "unit": "proofs/s", "cores": threads,
                     "sample": f"the same oracle proving proofs 0..{P-1} from the same witnesses in {t_p:.2f}s; c1 equal to GPU: {same_p}"},
           "single_thread": {"verifies_per_s": 1e3 / one["verify_ms"], "proofs_per_s": 1e3 / one["prove_ms"], "cores": 1,
                             "sample": "one proof of the batch proved and verified on one thread (BASELINE configs[0], benches/all.rs:55-71)"},
           "configs[0] one RangeProofNi, n=2048, CPU reference path": lat}
    return cpu, same and same_p


def pcie_leg(ctx, pb, expect, np, B):
    """the same verify step with HOST buffers (what a Rust caller of the crate hands over): H2D staging of ~1.5 GiB included"""
    host = pb.to(None)
    v = np.zeros(B, np.uint8)
    ctx.range_ni_verify(host.struct(), v, device=False)          # warm the ctx's staging blocks
    t0 = time.perf_counter()
    ctx.range_ni_verify(host.struct(), v, device=False)
    dt = time.perf_counter() - t0
    same = bool(np.array_equal(v, expect.cpu().numpy()))
    nbytes = sum(getattr(host, f).nbytes for f in ("n", "range", "ciphertext", "c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"))
    return {"value": B / dt, "unit": "verifies/s", "ms_per_step": 1e3 * dt, "host_bytes_staged": nbytes,
            "note": "zkp_range_ni_verify_batch on pageable host buffers, second call (staging blocks warm); never part of `value`"}, same


def other_configs(args, ctx, synth, torch, dev, sync, pb, wt, rank, enc_roofline, lpl, np):
    """legs for the other BASELINE.json configurations (rank-local, per GPU; not part of `value`)"""
    other = {}
    ok = True
    g = torch.Generator(device=dev); g.manual_seed(99 + rank)

    def rnd(shape):
        return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)

    def timed_once(fn):
        sync(); ctx.timing_reset(True); t0 = time.perf_counter()
        fn(); sync()
        dt = time.perf_counter() - t0
        kms, launches, me = ctx.timing_get(); ctx.timing_reset(False)
        return dt, kms, launches, me

    # configs[0]: the reference's own bench shape (benches/all.rs:55-77): ONE proof under the fixture key, host buffers in
    # and out (what a caller of the crate sees: staging and PCIe included); prove and verify timed separately
    pb1 = pb.slice(1, 2).to(None); wt1 = wt.slice(1, 2).to(None)
    v1 = np.zeros(1, np.uint8)

    def one_proof():
        ctx.range_ni_prove(pb1.struct(), wt1.struct(), None, None, None, device=False)      # warm-up
        t0 = time.perf_counter()
        ctx.range_ni_prove(pb1.struct(), wt1.struct(), None, None, None, device=False)
        t1 = time.perf_counter()
        ctx.range_ni_verify(pb1.struct(), v1, device=False)
        t2 = time.perf_counter()
        return {"prove_ms": 1e3 * (t1 - t0), "verify_ms": 1e3 * (t2 - t1), "prove_plus_verify_ms": 1e3 * (t2 - t0), "accepted": bool(v1[0] == 1),
                "limbs_per_lane": ctx.last_geometry()}

    ctx.set_geometry(0)                      # automatic geometry: a call this small runs on the latency engine (W = 9) when it is loaded
    try:
        rec0 = one_proof()
    finally:
        ctx.set_geometry(lpl)                # the batch legs are pinned to the throughput engine
    rec0["on_the_throughput_engine"] = one_proof()
    ok = ok and rec0["accepted"] and rec0["on_the_throughput_engine"]["accepted"]
    other["configs[0] one RangeProofNi, n=2048, host buffers (GPU latency)"] = rec0

    # configs[3]: 65536 NiCorrectKeyProof verifies, n = 2048, 65536 distinct (pseudo-)moduli: pure throughput shape,
    # every record is expected to be rejected (random sigma); accept parity is covered by tests/test_gpu_fullsize.py
    Bk, kwk = 65536, 64
    nk = rnd((Bk, kwk)); nk[:, 0] |= 1; nk[:, -1] |= -2**31
    sg = rnd((Bk, 11, kwk)); sg[:, :, -1] &= 0x3FFFFFFF
    vk = torch.full((Bk,), 9, dtype=torch.uint8, device=dev)
    ctx.correct_key_ni_verify(2048, Bk, nk, sg, b"KZen", vk); sync()      # warm-up
    dtk, kms_k, _, me_k = timed_once(lambda: ctx.correct_key_ni_verify(2048, Bk, nk, sg, b"KZen", vk))
    ach_k = me_k * modexp_limb_macs(2048, 2048) / (kms_k * 1e-3)
    other["configs[3] NiCorrectKeyProof verify, n=2048, batch=65536 distinct moduli (per GPU)"] = {
        "verifies_per_s": Bk / dtk, "modexp_per_s": me_k / (kms_k * 1e-3), "all_rejected_as_expected": bool((vk == 0).all().item()),
        "kernel": f"k_ck_check<{72 // lpl}>", "kernel_ms": kms_k, "achieved_limb_mac_per_s": ach_k, "frac": ach_k / PEAK_LIMB_MAC_PER_S,
        "frac_vs_16ms_kernel_peak": ach_k / PEAK_LIMB_MAC_PER_S_16MS_KERNELS,
        "traffic": (lambda per: per[0] * me_k if per[0] else None)(pmc_traffic_per_modexp(f"k_ck_check<{72 // lpl}>"))}
    ok = ok and bool((vk == 0).all().item())
    del nk, sg, vk

    def range_leg(nkey, nb, Bx, seed, kernel):
        """prove then verify of Bx proofs (one launch sequence each, no warm-up: the launches are seconds long)"""
        pbx, wtx = synth.synth_range_inputs(nkey, nb, Bx, seed=seed, device=dev)
        sync()
        ctx.paillier_enc(nb, Bx, pbx.n, 0 if isinstance(nkey, int) else nb // 32, wtx.x, wtx.r, pbx.ciphertext); sync()
        vx = torch.full((Bx,), 9, dtype=torch.uint8, device=dev)
        dtp, kp, lp, mp = timed_once(lambda: ctx.range_ni_prove(pbx.struct(), wtx.struct(), None, None, None, device=True))
        bad = torch.arange(0, Bx, 64, device=dev)
        pbx.resp_r1[bad, 0, 0] ^= 1
        exp = torch.ones(Bx, dtype=torch.uint8, device=dev); exp[bad] = 0
        dtv, kv, lv, mv = timed_once(lambda: ctx.range_ni_verify(pbx.struct(), vx, device=True))
        good = bool(torch.equal(vx, exp))
        rec = {"batch": Bx, "proofs_per_s": Bx / dtp, "verifies_per_s": Bx / dtv, "prove_ms": 1e3 * dtp, "verify_ms": 1e3 * dtv, "verdicts_ok": good,
               "prove_frac": mp * enc_limb_macs(nb) / (kp * 1e-3) / PEAK_LIMB_MAC_PER_S if kp else None,
               "roofline": enc_roofline(kv, lv, mv, nb, kernel)}
        del pbx, wtx, vx
        torch.cuda.empty_cache()
        return rec, good

    # SURVEY §8(d) config 3 "4096 distinct eks": every proof under its own 2048-bit key (fixed 5-bit windows: the exponent differs per item)
    if args.distinct_batch > 0:
        keys = synth.distinct_keys_2048(args.distinct_batch)
        rec, good = range_leg(keys, 2048, args.distinct_batch, 777 + rank, f"k_enc<{144 // lpl}, false> (per-proof keys: fixed-window ladder)")
        other[f"configs[2]/[1] with {args.distinct_batch} DISTINCT 2048-bit keys (products of pooled 1024-bit primes), prove + verify (per GPU)"] = rec
        ok = ok and good
    # configs[4]: RangeProofNi prove + verify at n = 4096 (8192-bit n^2) under a real 4096-bit key
    if args.big_batch > 0:
        n5 = synth.bench_key_4096()[2]
        rec, good = range_leg(n5, 4096, args.big_batch, 4321 + rank, f"k_enc<{288 // lpl}, true> (n = 4096: {288 // lpl} lanes x {lpl} limbs per 8192-bit integer)")
        other[f"configs[4] RangeProofNi prove+verify, n=4096 (4096-bit key p*q of bench_keys.json), batch={args.big_batch} (per GPU)"] = rec
        ok = ok and good
    return other, ok


def run_pmc_shape(args, ctx, synth, torch, dev, sync):
    """ONE dominant-kernel launch of a fixed, small shape; prints the modexp count of that launch (stdout, JSON)"""
    shape = args.pmc_shape
    if shape in ("enc2048", "enc2048keys", "enc4096"):
        nb = 4096 if shape == "enc4096" else 2048
        Bx = 128 if shape == "enc4096" else 512
        nkey = synth.bench_key_4096()[2] if shape == "enc4096" else (synth.distinct_keys_2048(Bx) if shape == "enc2048keys" else synth.BENCH_N)
        pbx, wtx = synth.synth_range_inputs(nkey, nb, Bx, seed=5, device=dev)
        sync()
        ctx.paillier_enc(nb, Bx, pbx.n, 0 if isinstance(nkey, int) else nb // 32, wtx.x, wtx.r, pbx.ciphertext); sync()
        ctx.range_ni_prove(pbx.struct(), wtx.struct(), None, None, None, device=True); sync()
        v = torch.zeros(Bx, dtype=torch.uint8, device=dev)
        ctx.timing_reset(True)
        ctx.range_ni_verify(pbx.struct(), v, device=True); sync()
        kms, launches, me = ctx.timing_get()
        print(json.dumps({"pmc_shape": shape, "verify_launch_modexps": me, "verify_launch_ms": kms, "all_modexps_of_the_kernel": me + Bx + 2 * Bx * 128,
                          "all_accepted": bool((v == 1).all().item())}))
    else:
        Bk, kwk = 8192, 64
        g = torch.Generator(device=dev); g.manual_seed(7)
        nk = torch.randint(-2**31, 2**31 - 1, (Bk, kwk), dtype=torch.int32, device=dev, generator=g); nk[:, 0] |= 1; nk[:, -1] |= -2**31
        sg = torch.randint(-2**31, 2**31 - 1, (Bk, 11, kwk), dtype=torch.int32, device=dev, generator=g); sg[:, :, -1] &= 0x3FFFFFFF
        vk = torch.zeros(Bk, dtype=torch.uint8, device=dev)
        ctx.timing_reset(True)
        ctx.correct_key_ni_verify(2048, Bk, nk, sg, b"KZen", vk); sync()
        kms, launches, me = ctx.timing_get()
        print(json.dumps({"pmc_shape": shape, "all_modexps_of_the_kernel": me, "launch_ms": kms}))


if __name__ == "__main__":
    main()
