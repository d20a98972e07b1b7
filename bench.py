#!/usr/bin/env python3
"""bench.py — RangeProofNi verify (headline) and prove throughput at n=2048, batch=4096 proofs per GPU.

A "step" of the headline metric is one pass of zkp_range_ni_verify_batch over one batch of
B synthetic proofs already resident in HBM (BASELINE.json configs[1]); the prove leg
(configs[2]) is timed the same way and reported beside it.  One process per GPU; for N>1 the
proof indices are sharded by rank (weak scaling: B proofs per rank), no collective on the
data path, and one RCCL all-gather reassembles the verdict vector (and the ciphertext slabs
of the prove leg) inside the timed region.

Prints ONE JSON line on rank 0.  See DESIGN.md §6 for the definitions of roofline/cpu_baseline."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# measured on MI355X (profiles/valu_rates_long_r01.jsonl, 16 ms kernels so that the clock has settled):
# v_mad_u64_u32, 16 independent accumulators, 8 waves/SIMD, 256 CUs -> 3.474e13 lane-MAC/s (4.53 cycles per wave64 issue);
# the short (1 ms) run of profiles/valu_rates_r01.jsonl read 3.19e13 and was used in the first bench lines of this round
PEAK_LIMB_MAC_PER_S = 3.474e13
HBM_PEAK_GBS = 8000.0
# HBM-side bytes per Enc of k_enc measured with rocprofv3 PMC passes of this same command
# (profiles/r01_pmc_bench_b512_v5kernel.json: FETCH_SIZE 86.2 KB + WRITE_SIZE 21.8 KB per Enc, raw counters;
# almost all of it is the per-exponentiation window table spilling out of L2, not operand traffic)
PMC_HBM_BYTES_PER_ENC = 86193.1 + 21814.2
# algorithmic 32x32->64 limb-MACs of one Enc at n=2048: 1.2*2048 modmuls x (2*128^2+128)
def enc_limb_macs(n_bits):
    Lw = 2 * n_bits // 32
    return 1.2 * n_bits * (2 * Lw * Lw + Lw)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="proofs per GPU")
    ap.add_argument("--n-bits", type=int, default=2048)
    ap.add_argument("--cpu-sample", type=int, default=96, help="proofs verified by the CPU baseline (0 = skip)")
    ap.add_argument("--no-prove-leg", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs for BASELINE.json configs[3] and configs[4]")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    zkp = importlib.import_module("zk-paillier_amd")
    from importlib import import_module
    synth = import_module("zk-paillier_amd.synth")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (RANK / WORLD_SIZE set) the RCCL path is exercised even at world size 1
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ctx = zkp.Context(local_rank)          # raises if the HIP library / a gfx950 GPU is missing
    lpl = int(zkp.load().zkp_build_limbs_per_lane())

    B, n_bits, EF = args.batch, args.n_bits, 128
    kw = n_bits // 32
    n = synth.BENCH_N
    assert n_bits == 2048, "the bench key is the reference's 2048-bit fixture"

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    # ---- inputs (untimed): witnesses in HBM, ciphertext = Enc(x, r) by the engine itself
    pb, wt = synth.synth_range_inputs(n, n_bits, B, seed=1234 + rank, device=dev)
    sync()
    ctx.paillier_enc(n_bits, B, pb.n, 0, wt.x, wt.r, pb.ciphertext)
    sync()
    pstruct, wstruct = pb.struct(), wt.struct()
    verdict = torch.zeros(B, dtype=torch.uint8, device=dev)
    gathered_v = torch.zeros(B * world, dtype=torch.uint8, device=dev) if use_dist else None
    gathered_c = None
    if use_dist and not args.no_prove_leg:
        gathered_c = [torch.empty((B * world,) + tuple(pb.c1.shape[1:]), dtype=pb.c1.dtype, device=dev) for _ in range(2)]

    def prove_step():
        ctx.range_ni_prove(pstruct, wstruct, None, None, None, device=True)
        if use_dist:
            ctx.synchronize()
            dist.all_gather_into_tensor(gathered_c[0], pb.c1)
            dist.all_gather_into_tensor(gathered_c[1], pb.c2)
            torch.cuda.synchronize()      # the next step overwrites c1/c2 on the engine's own stream

    def verify_step():
        ctx.range_ni_verify(pstruct, verdict, device=True)
        if use_dist:
            ctx.synchronize()
            dist.all_gather_into_tensor(gathered_v, verdict)
            torch.cuda.synchronize()

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
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms, launches, modexps

    # ---- prove leg (also produces the proofs the verify leg consumes)
    prove = None
    if args.no_prove_leg:
        prove_step(); sync()
    else:
        dt, kms, launches, modexps = timed(prove_step, args.steps, args.warmup)
        prove = {"value": B * world * args.steps / dt, "unit": "proofs/s", "ms_per_step": 1e3 * dt / args.steps,
                 "enc_kernel_ms_per_launch": kms / max(launches, 1), "launches": launches,
                 "achieved_limb_mac_per_s": modexps * enc_limb_macs(n_bits) / (kms * 1e-3) if kms else None}
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
    if use_dist:
        ok = ok and bool(torch.equal(gathered_v.view(world, B)[rank], expect))
        if gathered_c is not None:
            ok = ok and bool(torch.equal(gathered_c[0].view(world, B, *pb.c1.shape[1:])[rank], pb.c1))
    value = B * world * args.steps / dt
    enc_per_launch = modexps / max(launches, 1)
    ach = modexps * enc_limb_macs(n_bits) / (kms * 1e-3) if kms else 0.0
    # algorithmic bytes per Enc-check: r, m (kw words each) + expected ciphertext (2kw) + 8 B item
    bytes_per_enc = 4 * kw * 4 + 8
    roofline = {"bound": "valu", "achieved": ach / 1e12, "peak": PEAK_LIMB_MAC_PER_S / 1e12, "unit": "Tlimb-MAC/s",
                "frac": ach / PEAK_LIMB_MAC_PER_S, "traffic": PMC_HBM_BYTES_PER_ENC * enc_per_launch,
                "traffic_note": "bytes per launch = PMC-measured FETCH_SIZE+WRITE_SIZE per Enc (profiles/r01_pmc_bench_b512_v5kernel.json) x Enc of the launch; algorithmic operand bytes are ~1 KB per Enc",
                "kernel": f"k_enc<{144 // lpl}> (fused Enc-and-compare; {144 // lpl} lanes x {lpl} limbs per 4096-bit integer)",
                "kernel_ms_per_launch": kms / max(launches, 1),
                "modexps_per_launch": enc_per_launch,
                "hbm": {"achieved": modexps * bytes_per_enc / (kms * 1e-3) / 1e9 if kms else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s"}}

    # ---- CPU baseline: the C/GMP oracle on a bounded sample of the same proofs (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib   # test infrastructure: used here only as the reported CPU baseline / checker
        oracle = oracle_lib.Oracle()
        S = min(args.cpu_sample, B)
        # a sample that contains a tampered proof: proofs 0..S-1 (proof 0 is tampered)
        host = pb.slice(0, S).to(None)
        threads = usable_cores(oracle.max_threads())
        oracle.set_threads(threads)
        vo = np.zeros(S, np.uint8)
        t0 = time.perf_counter()
        oracle.range_ni_verify(host.struct(), vo)
        t_cpu = time.perf_counter() - t0
        same = bool(np.array_equal(vo, verdict[:S].cpu().numpy()))
        ok = ok and same
        cpu = {"value": S / t_cpu, "unit": "verifies/s", "cores": threads, "kind": "port",
               "sample": f"oracle (C + GMP 6.2.1 mpz_powm, OpenMP over (proof,row)) verifying proofs 0..{S-1} of the same batch in {t_cpu:.2f}s; verdicts equal to GPU: {same}"}

    # ---- short legs for the other BASELINE.json configurations (rank-local, reported per GPU; not part of `value`)
    other = None
    if not args.no_other_configs and world == 1:      # single-GPU shapes: reported at N=1 only
        other = {}
        g = torch.Generator(device=dev); g.manual_seed(99 + rank)
        def rnd(shape):
            return torch.randint(-2**31, 2**31 - 1, shape, dtype=torch.int32, device=dev, generator=g)
        # configs[0]: the reference's own bench shape (benches/all.rs:55-77): ONE proof under the fixture key, host buffers in
        # and out (what a caller of the crate sees: staging and PCIe included); prove and verify timed separately
        pb1 = pb.slice(0, 1).to(None); wt1 = wt.slice(0, 1).to(None)
        if True:
            v1 = np.zeros(1, np.uint8)
            ctx.range_ni_prove(pb1.struct(), wt1.struct(), None, None, None, device=False)      # warm-up
            t0 = time.perf_counter()
            ctx.range_ni_prove(pb1.struct(), wt1.struct(), None, None, None, device=False)
            t1 = time.perf_counter()
            ctx.range_ni_verify(pb1.struct(), v1, device=False)
            t2 = time.perf_counter()
            ok = ok and bool(v1[0] == 1)
            other["configs[0] one RangeProofNi, n=2048, host buffers (latency)"] = {
                "prove_ms": 1e3 * (t1 - t0), "verify_ms": 1e3 * (t2 - t1), "prove_plus_verify_ms": 1e3 * (t2 - t0), "accepted": bool(v1[0] == 1)}
        # configs[3]: 65536 NiCorrectKeyProof verifies, n = 2048, 65536 distinct (pseudo-)moduli: pure throughput shape,
        # every record is expected to be rejected (random sigma); accept parity is covered by tests/test_gpu_fullsize.py
        Bk, kwk = 65536, 64
        nk = rnd((Bk, kwk)); nk[:, 0] |= 1; nk[:, -1] |= -2**31
        sg = rnd((Bk, 11, kwk)); sg[:, :, -1] &= 0x3FFFFFFF
        vk = torch.full((Bk,), 9, dtype=torch.uint8, device=dev)
        sync()
        ctx.correct_key_ni_verify(2048, Bk, nk, sg, b"KZen", vk); sync()      # warm-up
        ctx.timing_reset(True); t0 = time.perf_counter()
        ctx.correct_key_ni_verify(2048, Bk, nk, sg, b"KZen", vk); sync()
        dtk = time.perf_counter() - t0
        kms_k, _, me_k = ctx.timing_get(); ctx.timing_reset(False)
        other["configs[3] NiCorrectKeyProof verify, n=2048, batch=65536 distinct moduli (per GPU)"] = {
            "verifies_per_s": Bk / dtk, "modexp_per_s": me_k / (kms_k * 1e-3), "all_rejected_as_expected": bool((vk == 0).all().item()),
            "achieved_limb_mac_per_s": me_k * 1.2 * 2048 * (2 * 64 * 64 + 64) / (kms_k * 1e-3)}
        del nk, sg, vk
        # configs[4]: RangeProofNi prove + verify at n = 4096 (8192-bit n^2); batch reduced to 256 proofs per GPU to stay short
        B5, nb5 = 256, 4096
        n5 = (1 << 4095) | int.from_bytes(os.urandom(500), "big") | 1          # odd 4096-bit pseudo-modulus: prove -> verify round trip is key-agnostic
        pb5, wt5 = synth.synth_range_inputs(n5, nb5, B5, seed=4321 + rank, device=dev)
        sync()
        ctx.paillier_enc(nb5, B5, pb5.n, 0, wt5.x, wt5.r, pb5.ciphertext); sync()
        v5 = torch.full((B5,), 9, dtype=torch.uint8, device=dev)
        t0 = time.perf_counter()
        ctx.range_ni_prove(pb5.struct(), wt5.struct(), None, None, None, device=True); sync()
        t1 = time.perf_counter()
        ctx.range_ni_verify(pb5.struct(), v5, device=True); sync()
        t2 = time.perf_counter()
        ok5 = bool((v5 == 1).all().item())
        ok = ok and ok5
        other["configs[4] RangeProofNi prove+verify, n=4096, batch=256 (per GPU, reduced from 4096)"] = {
            "proofs_per_s": B5 / (t1 - t0), "verifies_per_s": B5 / (t2 - t1), "all_accepted": ok5}
        del pb5, wt5

    if rank == 0:
        out = {"metric": "RangeProofNi proofs/sec + verifies/sec, n=2048, batch=4096 per GPU (value = verifies/sec; proofs/sec in prove.value)", "value": value, "unit": "verifies/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (29-bit limbs, u64 accumulate)",
               "data": "synthetic", "verdicts_ok": ok,
               "config": {"workload": f"BASELINE.json configs[1]: batch={B} RangeProofNi verify per GPU, n={n_bits} (reference fixture key), "
                                      f"128 rows/proof, 1/64 of the proofs tampered; prove leg = configs[2]",
                          "parallelism": f"proof-index sharding x{world}, RCCL all-gather of verdicts (verify) and c1/c2 slabs (prove)" if use_dist else "single GPU"},
               "prove": prove, "roofline": roofline, "cpu_baseline": cpu, "other_configs": other}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
