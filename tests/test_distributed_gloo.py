"""CPU test of the N>1 path: world_size 2 over gloo.  Each rank verifies its contiguous block of a
proof batch (the per-rank compute is the ORACLE here — there is no GPU — the plumbing under test is
zk-paillier_amd/shard.py: index partition + the single all-gather) and the gathered verdict vector must
equal the single-process result.  The second test drives bench.py's own step functions (bench.make_steps:
prove / verify on the rank's block + the gather through shard.py) with an oracle-backed engine, i.e. the code
`bench.py --gpus N` times, at world size 2."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers as H
from helpers import zkp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_bits, B, ret):
    import importlib
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    shard = importlib.import_module("zk-paillier_amd.shard")
    oracle = oracle_lib.Oracle()
    n = H.test_key(512)[2]
    cases = H.build_range_case(b"gloo", [n], n_bits, B)
    cases[B - 1] = H.build_range_case(b"gloo-bad", [n], n_bits, 1, honest=False)[0]
    pb, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None)

    def verify(local):
        v = np.zeros(local.batch, np.uint8)
        oracle.range_ni_verify(local.struct(), v)
        return torch.from_numpy(v)

    out = shard.sharded_verify(verify, B, world, rank, lambda lo, hi: pb.slice(lo, hi))
    # prove-side slab gather (unequal row counts)
    lo, hi = shard.shard_range(B, world, rank)
    counts = [shard.shard_range(B, world, r)[1] - shard.shard_range(B, world, r)[0] for r in range(world)]
    c1 = shard.all_gather_slabs(torch.from_numpy(pb.c1[lo:hi].view(np.int32)), world, counts)
    ok_c1 = bool(np.array_equal(c1.numpy().view(np.uint32), pb.c1))
    # BASELINE configs[3]: NiCorrectKeyProof verification sharded by key index through the same helper
    from helpers import pm, L
    keys = [H.test_key(512, tag=t) for t in range(5)]
    n_arr = L.ints_to_limbs([k[2] for k in keys], 32)
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(k[0], k[1], b"KZen"), 32) for k in keys])
    sig[3, 7, 0] ^= 1                                    # one tampered proof
    ck = shard.sharded_verify(lambda sl: torch.from_numpy(oracle.correct_key_ni_verify(1024, n_arr[sl], np.ascontiguousarray(sig[sl]), b"KZen")),
                              len(keys), world, rank, lambda lo, hi: slice(lo, hi))
    if rank == 0:
        full = np.zeros(B, np.uint8)
        oracle.range_ni_verify(pb.struct(), full)
        ck_full = oracle.correct_key_ni_verify(1024, n_arr, sig, b"KZen")
        ret.put((out.numpy().tolist(), full.tolist(), ok_c1 and ck.numpy().tolist() == ck_full.tolist() == [1, 1, 1, 0, 1]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    shard = __import__("importlib").import_module("zk-paillier_amd.shard")
    for total in (0, 1, 5, 4096, 65536 + 3):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_range(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo_verify_gather():
    import torch.multiprocessing as mp
    world, B, n_bits = 2, 3, 1024
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_bits, B, ret)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, full, ok_c1 = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert gathered == full == [zkp.VERDICT_ACCEPT, zkp.VERDICT_ACCEPT, zkp.VERDICT_REJECT]
    assert ok_c1


class _OracleEngine:
    """stands in for bench.GpuEngine on a CPU box: same interface, per-rank compute by the oracle on torch CPU tensors"""

    def __init__(self, oracle):
        self.oracle = oracle

    def prove(self, pb, wt):
        assert self.oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None) == 0

    def verify(self, pb, verdict):
        assert self.oracle.range_ni_verify(pb.struct(), verdict.numpy()) == 0

    def correct_key_verify(self, n_bits, n, sigma, salt, verdict):
        verdict.numpy()[:] = self.oracle.correct_key_ni_verify(n_bits, n.numpy().view(np.uint32), np.ascontiguousarray(sigma.numpy().view(np.uint32)), salt)

    def before_collective(self):
        pass

    def after_collective(self):
        pass


def _rank_batch(rank, n_bits, B, oracle):
    n = H.test_key(512)[2]
    cases = H.build_range_case(b"bench-steps-%d" % rank, [n], n_bits, B)
    if rank == 1:
        cases[0] = H.build_range_case(b"bench-steps-bad", [n], n_bits, 1, honest=False)[0]
    pb, wt = H.fill_batch(cases, n_bits, True, oracle)
    return pb.to("cpu"), wt.to("cpu")


def _bench_worker(rank, world, port, n_bits, B, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    sys.path.insert(0, H.ROOT)
    import bench
    oracle = oracle_lib.Oracle()
    pb, wt = _rank_batch(rank, n_bits, B, oracle)
    verdict = torch.zeros(B, dtype=torch.uint8)
    # --gather verdicts: a prove step exchanges nothing, the verify step still gathers its verdict bytes; receive buffers are
    # allocated by make_steps, once (their sizes are what bench.py prints per rank)
    p2, v2, out2 = bench.make_steps(_OracleEngine(oracle), pb, wt, verdict, world, None, "verdicts")
    p2(); v2()
    verdicts_mode = ("c1" not in out2, out2["verdict"].tolist(), out2["recv_bytes"])
    prove_step, verify_step, out = bench.make_steps(_OracleEngine(oracle), pb, wt, verdict, world)
    recv = out["recv_bytes"]
    assert recv == {"prove": 2 * world * B * 128 * (2 * n_bits // 32) * 4, "verify": world * B} and verdicts_mode[2] == {"prove": 0, "verify": world * B}
    prove_step()
    first_buffer = out["c1"].data_ptr()
    verify_step()
    prove_step()                                    # a second step reuses the receive buffers
    assert out["c1"].data_ptr() == first_buffer
    assert verdicts_mode[0] and verdicts_mode[1] == out["verdict"].tolist()
    if rank == 0:
        # single-process reference: both ranks' batches proved and verified here
        exp_v, exp_c1 = [], []
        for r in range(world):
            qb, qw = _rank_batch(r, n_bits, B, oracle)
            oracle.range_ni_prove(qb.struct(), qw.struct(), None, None, None)
            v = np.zeros(B, np.uint8)
            oracle.range_ni_verify(qb.struct(), v)
            exp_v += v.tolist(); exp_c1.append(qb.c1)
        ret.put((out["verdict"].tolist(), exp_v, bool(torch.equal(out["c1"], torch.cat(exp_c1, dim=0))), list(out["c2"].shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_functions_world_size_2_gloo():
    import torch.multiprocessing as mp
    world, B, n_bits = 2, 2, 1024
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, n_bits, B, ret)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, expected, c1_ok, c2_shape = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert gathered == expected == [zkp.VERDICT_ACCEPT, zkp.VERDICT_ACCEPT, zkp.VERDICT_REJECT, zkp.VERDICT_ACCEPT]
    assert c1_ok and c2_shape == [world * B, 128, 2 * n_bits // 32]


def _sharded_worker(rank, world, port, ret):
    """the sharded legs of `bench.py --gpus N` (strong scaling): BASELINE configs[3] — NiCorrectKeyProof keys cut into blocks,
    bench.make_correct_key_step — and a RangeProofNi batch of a size the world does not divide (bench.make_steps with counts:
    configs[4]'s shape), per-rank compute by the oracle, gathered outputs against the single-process run."""
    import importlib
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from helpers import pm, L
    sys.path.insert(0, H.ROOT)
    import bench
    shard = importlib.import_module("zk-paillier_amd.shard")
    oracle = oracle_lib.Oracle()
    eng = _OracleEngine(oracle)
    # configs[3]: 5 keys in all (unequal blocks 3 + 2), one tampered proof
    keys = [H.test_key(512, tag=t) for t in range(5)]
    n_arr = L.ints_to_limbs([k[2] for k in keys], 32)
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(k[0], k[1], b"KZen"), 32) for k in keys])
    sig[3, 7, 0] ^= 1
    lo, hi = shard.shard_range(len(keys), world, rank)
    counts = bench.block_counts(len(keys), world)
    vk = torch.full((hi - lo,), 9, dtype=torch.uint8)
    step, out = bench.make_correct_key_step(eng, 1024, torch.from_numpy(n_arr[lo:hi].view(np.int32)), torch.from_numpy(np.ascontiguousarray(sig[lo:hi]).view(np.int32)),
                                            b"KZen", vk, world, counts)
    step()
    ck = out["verdict"].tolist()
    # configs[4]'s shape: 3 proofs in all (blocks 2 + 1), prove + verify with the gathers
    n = H.test_key(512)[2]
    cases = H.build_range_case(b"sharded-legs", [n], 1024, 3)
    cases[2] = H.build_range_case(b"sharded-legs-bad", [n], 1024, 1, honest=False)[0]
    pb, wt = H.fill_batch(cases, 1024, True, oracle)
    pb, wt = pb.to("cpu"), wt.to("cpu")
    plo, phi = shard.shard_range(3, world, rank)
    loc, wloc = pb.slice(plo, phi), wt.slice(plo, phi)
    v = torch.zeros(phi - plo, dtype=torch.uint8)
    p_step, v_step, got = bench.make_steps(eng, loc, wloc, v, world, bench.block_counts(3, world))
    p_step(); v_step()
    if rank == 0:
        ck_full = oracle.correct_key_ni_verify(1024, n_arr, sig, b"KZen").tolist()
        oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None)      # (rank 0's block is already proved in place: same bytes again)
        full = np.zeros(3, np.uint8)
        oracle.range_ni_verify(pb.struct(), full)
        ret.put((ck, ck_full, got["verdict"].tolist(), full.tolist(), bool(torch.equal(got["c1"], pb.c1)), list(got["c2"].shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_bench_legs_world_size_2_gloo():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    ck, ck_full, verdicts, full, c1_ok, c2_shape = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ck == ck_full == [1, 1, 1, 0, 1]
    assert verdicts == full == [zkp.VERDICT_ACCEPT, zkp.VERDICT_ACCEPT, zkp.VERDICT_REJECT]
    assert c1_ok and c2_shape == [3, 128, 64]


class _OracleCtx:
    """what bench.scaling_leg asks of a ctx, on the oracle"""

    def __init__(self, oracle):
        self.oracle = oracle

    def paillier_enc(self, n_bits, count, n, n_stride, m, r, out):
        u = lambda t: np.ascontiguousarray(t.numpy().view(np.uint32))
        res = self.oracle.paillier_enc(n_bits, u(n).reshape(-1), n_stride, u(m), u(r))
        out.copy_(__import__("torch").from_numpy(res.view(np.int32)).view(out.shape))


def _scaling_worker(rank, world, port, ret):
    """both scaling modes of `bench.py --gpus 2` through bench.scaling_leg — the code the driver's SCALE run executes — with the oracle as
    the engine, and the part of the JSON line that makes such a run self-validating"""
    import argparse
    import importlib
    import json
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    sys.path.insert(0, H.ROOT)
    import bench
    shard = importlib.import_module("zk-paillier_amd.shard")
    synth = importlib.import_module("zk-paillier_amd.synth")
    oracle = oracle_lib.Oracle()
    oracle.set_threads(max(1, min(8, oracle.max_threads() // world)))
    args = argparse.Namespace(batch=2, steps=1, warmup=0, gather="all")

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        dist.barrier()
        return time.perf_counter() - t0, 0.0, 0, 0

    values, ok = {}, True
    for kind in ("weak", "strong"):
        values[kind], same = bench.scaling_leg(kind, args, _OracleEngine(oracle), synth, shard, torch, torch.device("cpu"), _OracleCtx(oracle), lambda: None, timed,
                                               synth.BENCH_N, 2048, world, rank)
        ok = ok and same
    mine = dict(bench.gpu_identity(0, rank))
    gpus = [None] * world
    dist.all_gather_object(gpus, mine)
    if rank == 0:
        line = json.dumps({"n_gpus": world, "scaling": "weak", "value": values["weak"]["verifies_per_s"], "scaling_values": values,
                           "rccl": bench.rccl_block(dist.get_backend(), world, gpus), "verdicts_ok": ok})
        ret.put(line)
    dist.barrier()
    dist.destroy_process_group()


def test_a_two_rank_bench_line_carries_both_scaling_modes_and_says_who_answered():
    """round-5 verdict item 5: the line of `bench.py --gpus N` reports the STRONG-scaling value (BASELINE.json: "batch=4096, 1/2/4/8 GPU" —
    the batch IN ALL) beside the weak one, compute apart from gather, and how many ranks really took part"""
    import json
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scaling_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    line = json.loads(ret.get(timeout=900))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert line["verdicts_ok"] and line["rccl"]["ranks_seen"] == line["n_gpus"] == line["rccl"]["world_size"] == 2 and line["rccl"]["backend"] == "gloo"
    sv = line["scaling_values"]
    assert sv["weak"]["proofs_total"] == 4 and sv["weak"]["proofs_per_rank"] == 2
    assert sv["strong"]["proofs_total"] == 2 and sv["strong"]["proofs_per_rank"] == 1
    for kind in ("weak", "strong"):
        v = sv[kind]
        assert v["verifies_per_s"] > 0 and v["proofs_per_s"] > 0
        for ph in (v["verify_phases_rank0"], v["prove_phases_rank0"]):
            assert ph["steps"] == 1 and ph["compute_ms"] > 0 and ph["gather_ms"] >= 0
            assert ph["compute_ms"] + ph["gather_ms"] <= 1.05 * max(v["verify_ms_per_step"], v["prove_ms_per_step"]) + 50
    assert line["value"] == sv["weak"]["verifies_per_s"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`--gpus 8` under a 1-rank launcher must not print a 1-GPU line: exit status 2 before any GPU work"""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
