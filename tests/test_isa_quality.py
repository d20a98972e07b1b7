"""The numbers of DESIGN.md §8 depend on a property of the COMPILED kernels that no functional test sees: the fully unrolled product
loops of the ladders (1962 multiply-adds per block of a squaring, 2592 of a product) must not touch scratch memory — with two waves
per SIMD every spill inside them stalls the multiply-add pipe (round 3: the first squaring build kept ~30 scratch accesses per
product and gained +10 % instead of +24 %).  __graft_entry__.build() keeps the device assembly of the throughput engine under
build/v_isa/isa.s; this test reads it (and is skipped when the library was built another way)."""
import os
import re
import subprocess

import pytest

import helpers as H

ISA = os.path.join(H.ROOT, "build", "v_isa", "isa.s")


def hot_blocks(text, kernel):
    """[(multiply-adds, scratch accesses)] of the basic blocks of `kernel` that hold a whole product"""
    out = []
    for n in re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M):
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        if kernel not in dem:
            continue
        j = text.index("\n" + n + ":")
        body = text[j:text.index(".Lfunc_end", j)]
        for b in re.split(r"\n\.LBB\d+_\d+:", body):
            mads = b.count("v_mad_u64_u32")
            if mads >= 1900:
                out.append((mads, len(re.findall(r"scratch_(?:load|store)", b))))
    return out


@pytest.mark.skipif(not os.path.exists(ISA) or os.path.getmtime(ISA) + 900 < os.path.getmtime(H.zkp.LIB_PATH),
                    reason="no device assembly beside the built library (run __graft_entry__.build())")
def test_product_loops_of_the_ladders_are_free_of_scratch_accesses():
    text = open(ISA).read()
    for extra in ("isa_keys_enc.s", "isa_keys_ck.s", "isa_basen.s"):   # the per-item-exponent kernels (zkp_kernels_keys.hip), the base-n kernels (zkp_kernels_basen.hip)
        text += open(os.path.join(os.path.dirname(ISA), extra)).read()
    # (kernel, most scratch accesses tolerated in the squaring block, ... in the ladder's product block)
    for kernel, sq_max, mul_max in (("k_enc<4, true, false>", 0, 2), ("k_enc<4, false, false>", 0, 0), ("k_ck_check<2, false>", 0, 0),
                                    ("k_enc<8, true, false>", 0, 2), ("k_modexp<4, false, false, false>", 0, 0), ("k_modexp<2, false, false, false>", 0, 0)):
        blocks = hot_blocks(text, kernel)
        sq = [s for m, s in blocks if m == 1962]
        mul = [s for m, s in blocks if m == 2592]
        assert sq and mul, (kernel, blocks)                      # 54.5 and 72 multiply-adds x 36 sub-steps: the montsqr / montmul<ORUP> blocks
        assert min(sq) <= sq_max, (kernel, "squaring block", sq)
        assert min(mul) <= mul_max, (kernel, "product block", mul)
    # The base-n kernels (kernels_basen.hpp): the a side of a squaring (1962 multiply-adds per block + the 36 of the b side's initial columns,
    # which the compiler places in the same block) and TWO copies of the n-sized product body — the b side of the squarings on their own
    # path, and the three slots of every other base-n product.  The first builds of that file carried quotient digits and pending results
    # through these bodies in registers: 35 - 450 scratch accesses per block and 3.0 s instead of 1.2 s per verify step.
    # (the per-proof-keys variant reads C3 from global memory at the start of a b side: a few reloads around that load are tolerated)
    for kernel, sq_max in (("k_enc_basen<2>", 2), ("k_enc_basen<4>", 2), ("k_enc_basen_keys<2>", 12), ("k_enc_basen_keys<4>", 12)):
        blocks = hot_blocks(text, kernel)
        sq = [s for m, s in blocks if 1962 <= m <= 1998]
        mul = [s for m, s in blocks if m == 2592]
        assert len(sq) == 1 and len(mul) == 2, (kernel, blocks)      # more bodies than these cost instruction-cache room and registers
        assert sq[0] <= sq_max and max(mul) <= 2, (kernel, blocks)
