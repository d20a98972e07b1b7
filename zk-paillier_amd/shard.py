"""Multi-GPU plumbing: proofs are independent units, so a batch is cut into
contiguous blocks of proof indices, one block per rank/GPU; no collective on the data path.
The only exchange step is ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests)
that reassembles per-rank output slabs (verdict bytes; ciphertext slabs of a prove batch)."""


def shard_range(total: int, world: int, rank: int):
    """contiguous, balanced block of [0, total) owned by `rank` (first total % world ranks get one more)"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class GatherBuffers:
    """receive (and, for unequal blocks, padding) buffers of ONE recurring all-gather, allocated once outside the timed loop:
    at N = 8 the prove-side gather of a 4096-proof rank block receives 8 x 268 MB = 2.1 GB per output (c1, c2: 4.3 GB together) per rank, which a serving
    loop must not allocate per step.  nbytes = what this rank receives per gather."""

    def __init__(self, like, world: int, counts=None):
        import torch
        self.world, self.counts = world, (None if counts is None or len(set(counts)) == 1 else list(counts))
        rows = like.shape[0] if self.counts is None else max(self.counts)
        tail = tuple(like.shape[1:])
        self.pad = None if self.counts is None else torch.zeros((rows,) + tail, dtype=like.dtype, device=like.device)
        self.out = torch.empty((world * rows,) + tail, dtype=like.dtype, device=like.device)
        self.nbytes = self.out.numel() * self.out.element_size()


def all_gather_slabs(local, world: int, counts=None, buffers=None):
    """all-gather equally sized slabs along dim 0 -> tensor [world * local.shape[0], ...].
    With `counts` (rows per rank, unequal) slabs are padded to max(counts) and trimmed after the gather.
    buffers: a GatherBuffers made for this shape (then nothing is allocated here, except the trimmed copy of unequal blocks)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local            # no process group: nothing to exchange (with one, the degenerate gather still runs: same code at every N)
    if local.is_cuda and dist.get_backend() == "gloo":
        # device-resident slabs under the gloo backend (several ranks sharing one GPU in a functional check of the N > 1 path:
        # RCCL refuses that): the exchange goes through host memory
        return all_gather_slabs(local.cpu(), world, counts).to(local.device)
    if counts is not None and len(set(counts)) == 1:
        counts = None
    if counts is None:
        out = buffers.out if buffers is not None else torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    m = max(counts)
    pad = buffers.pad if buffers is not None else torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = buffers.out if buffers is not None else torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0)


def sharded_verify(verify_fn, batch_total: int, world: int, rank: int, make_local):
    """Run `verify_fn(local_batch) -> uint8 tensor[local]` on this rank's block and all-gather the verdicts.
    make_local(lo, hi) builds the rank-local batch view."""
    lo, hi = shard_range(batch_total, world, rank)
    local = verify_fn(make_local(lo, hi))
    counts = [shard_range(batch_total, world, r)[1] - shard_range(batch_total, world, r)[0] for r in range(world)]
    return all_gather_slabs(local, world, counts if len(set(counts)) > 1 else None)
