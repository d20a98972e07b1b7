"""Synthetic RangeProofNi workloads generated directly in HBM (torch is the device-memory
container; all values are little-endian 32-bit limb arrays stored as int32 bit patterns).

Distributions follow the reference's samplers in shape, not in RNG: range is a
256-bit value (benches/all.rs:79, range_proof_ni.rs:133), x < range/3 (range_proof_ni.rs:166),
w1 in [T, 2T), w2 = w1 - T, fair-coin swap (range_proof.rs:136-149), r, r1, r2 below n."""
import numpy as np

from . import limbs as L
from .batch import RangeBatch, RangeWitness

# The fixed keypair of the reference's tests/benches (range_proof_ni.rs:141-145, benches/all.rs:73-77): p * q
BENCH_P = 148677972634832330983979593310074301486537017973460461278300587514468301043894574906886127642530475786889672304776052879927627556769456140664043088700743909632312483413393134504352834240399191134336344285483935856491230340093391784574980688823380828143810804684752914935441384845195613674104960646037368551517
BENCH_Q = 158741574437007245654463598139927898730476924736461654463975966787719309357536545869203069369466212089132653564188443272208127277664424448947476335413293018778018615899291704693105620242763173357203898195318179150836424196645745308205164116144020613415407736216097185962171301808761138424668335445923774195463
BENCH_N = BENCH_P * BENCH_Q


def bench_key_4096():
    """(p, q, n): the 4096-bit Paillier modulus of BASELINE.json configs[4] (tools/make_bench_keys.py; the reference
    fixes only a 2048-bit keypair)"""
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_keys.json")))["key4096"]
    p, q = int(k["p"], 16), int(k["q"], 16)
    return p, q, p * q


def distinct_keys_2048(count: int):
    """`count` distinct 2048-bit RSA moduli: products p_i * p_j (i < j) of the 92 pooled 1024-bit primes of
    bench_keys.json ("4096 distinct eks")"""
    import json
    import os
    pool = [int(v, 16) for v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_keys.json")))["pool1024"]]
    out = []
    for j in range(1, len(pool)):
        for i in range(j):
            out.append(pool[i] * pool[j])
            if len(out) == count:
                return out
    raise ValueError(f"the prime pool yields {len(out)} moduli, {count} asked for")


def _rand_limbs(torch, gen, shape, device):
    v = torch.randint(0, 1 << 32, shape, dtype=torch.int64, device=device, generator=gen)
    return v


def _to_i32(torch, v):
    """int64 tensor holding values in [0, 2^32) -> int32 bit patterns"""
    return torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)


def synth_range_inputs(n, n_bits: int, batch: int, seed: int, device="cuda", ef: int = 128, range_bits: int = 256):
    """-> (RangeBatch with n/range filled [ciphertext still zero], RangeWitness), all on `device`.
    n: one int (shared key) or a list of `batch` ints (one key per proof, n_stride = kw)."""
    import torch
    kw = n_bits // 32
    rl = range_bits // 32
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    shared = isinstance(n, int)
    pb = RangeBatch(n_bits, batch, ef, shared_key=shared, device=device)
    wt = RangeWitness(n_bits, batch, ef, device=device)
    if shared:
        pb.n.copy_(torch.from_numpy(L.int_to_limbs(n, kw).view(np.int32)).to(device).view(1, kw))
    else:
        assert len(n) == batch
        pb.n.copy_(torch.from_numpy(L.ints_to_limbs(list(n), kw).view(np.int32)).to(device))
        n = min(n)
    # range: range_bits with the top bit set
    rng = _rand_limbs(torch, gen, (batch, rl), device)
    rng[:, rl - 1] |= 1 << 31
    pb.range[:, :rl] = _to_i32(torch, rng)
    # T = range // 3 on the host (B small integers), back as limbs
    rng_np = rng.cpu().numpy().astype(np.uint32)
    T = np.zeros((batch, rl), dtype=np.uint32)
    for b in range(batch):
        T[b] = L.int_to_limbs(L.limbs_to_int(rng_np[b]) // 3, rl)
    Tt = torch.from_numpy(T.astype(np.int64)).to(device)
    # x < 2^(range_bits-8) < T
    x = _rand_limbs(torch, gen, (batch, rl), device)
    x[:, rl - 1] &= 0x00FFFFFF
    wt.x[:, :rl] = _to_i32(torch, x)
    nbits_r = n.bit_length() - 8            # r, r1, r2 < 2^(bitlen(n)-8) < n
    def rand_below_n(shape_prefix):
        v = _rand_limbs(torch, gen, (*shape_prefix, kw), device)
        full, rem = nbits_r // 32, nbits_r % 32
        v[..., full] &= (1 << rem) - 1
        v[..., full + 1:] = 0
        return _to_i32(torch, v)
    wt.r.copy_(rand_below_n((batch,)))
    wt.r1.copy_(rand_below_n((batch, ef)))
    wt.r2.copy_(rand_below_n((batch, ef)))
    # u < 2^(range_bits-8) < T ;  hi = T + u in [T, 2T) ; lo = u ; fair-coin swap
    u = _rand_limbs(torch, gen, (batch, ef, rl), device)
    u[..., rl - 1] &= 0x00FFFFFF
    hi = torch.zeros_like(u)
    carry = torch.zeros((batch, ef), dtype=torch.int64, device=device)
    for k in range(rl):
        s = u[..., k] + Tt[:, k].unsqueeze(1) + carry
        hi[..., k] = s & 0xFFFFFFFF
        carry = s >> 32
    coin = torch.randint(0, 2, (batch, ef, 1), dtype=torch.int64, device=device, generator=gen).bool()
    w1 = torch.where(coin, u, hi)
    w2 = torch.where(coin, hi, u)
    wt.w1[..., :rl] = _to_i32(torch, w1)
    wt.w2[..., :rl] = _to_i32(torch, w2)
    return pb, wt
