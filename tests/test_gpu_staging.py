"""The ctx's cache of staging blocks (host-pointer calls) is bounded: a long-lived ctx that serves many batch shapes must not
pile up device memory (advisor finding, round 2: blocks were never evicted)."""
import numpy as np
import pytest
import torch

import helpers as H  # noqa: F401  (puts the repo root and tests/ on sys.path)
from helpers import pm, L

pytestmark = pytest.mark.gpu


def test_staging_cache_does_not_grow_with_the_number_of_shapes(zkp):
    ctx = zkp.Context(0)
    try:
        ctx.set_geometry(36)
        d = pm.Drbg(b"staging")
        mod = d.bits(2048) | 1 | (1 << 2047)
        m = L.ints_to_limbs([mod], 64)
        e = L.ints_to_limbs([5], 1)

        def call(count):
            b = np.full((count, 64), 3, np.uint32)
            out = np.zeros_like(b)
            ctx.modexp(2048, 32, count, b, e, 0, m, 0, out)            # host buffers: staged through the ctx's blocks

        call(60000); ctx.synchronize()                                  # (also sizes the ctx's own scratch and window table once)
        ctx.release_staging()
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        sizes = [20000 + 1100 * k for k in range(36)]                   # ascending: no cached block is ever large enough for the next call
        for c in sizes:
            call(c)
        ctx.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        unbounded = sum(2 * c * 64 * 4 for c in sizes)                  # what a cache without eviction would hold: ~720 MB
        assert unbounded > (600 << 20)
        # the bound is twice a call's own footprint (at least 64 MiB) + the blocks of the last call
        assert free0 - free1 < (200 << 20), (free0 - free1) / 2**20
        ctx.release_staging()
    finally:
        ctx.close()


def test_diag_table_traffic_moves_the_bytes_it_reports(zkp):
    """zkp_diag_table_traffic: the calibration aid of profiles/collect_pmc.sh (a known number of bytes in the ladders' table access
    pattern) — here only that it runs in both modes and reports resident groups x 32 entries x 576 B x passes"""
    ctx = zkp.Context(0)
    try:
        rd = ctx.diag_table_traffic(0, 2)
        wr = ctx.diag_table_traffic(1, 3)
        assert rd > 0 and rd % (32 * 576 * 2) == 0 and wr * 2 == rd * 3
        with pytest.raises(zkp.ZkpError):
            ctx.diag_table_traffic(2, 1)
    finally:
        ctx.close()
