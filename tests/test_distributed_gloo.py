"""CPU test of the N>1 path: world_size 2 over gloo.  Each rank verifies its contiguous block of a
proof batch (the per-rank compute is the ORACLE here — there is no GPU — the plumbing under test is
zk-paillier_amd/shard.py: index partition + the single all-gather) and the gathered verdict vector must
equal the single-process result."""
import os
import socket

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
