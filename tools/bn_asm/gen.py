"""Generator of the fixed-register base-n product engine of k_enc_basen<G> (gfx950, W = 36 limbs per lane, G = 2 / 4 lanes per n-sized
integer: n of 2048 / 4096 bits) — csrc/kernels_basen_asm_g2.inc, csrc/kernels_basen_asm_g4.inc.

Why assembler: the compiled product bodies of csrc/kernels_basen.hpp are clean (89 % multiply-adds, no
scratch access), but every attempt to change what surrounds them — ONE copy of the product body for all call sites, the a part of a
window multiplication kept in registers, the wide cross product (profiles/r05/experiments/README.md) — was lost to the compiler's register
allocation or to the 64 KB instruction cache two CUs share.  Here the register file is laid out by hand, the n-sized product exists
once, and the code between products (carry chains, the b side's initial columns, staging) is fused and chunked so that no 36-register
temporary exists at all.

What is generated (entry points, called from C++ through `s_swappc_b64` with pinned registers; csrc/kernels_basen.hpp: bn_asm_*):

  zkp_bn_sqr_run_g<G>   (a, b) staged in LDS  <-  (a, b)^(2^count)            [the squarings of the ladder: 79 % of an Enc]
  zkp_bn_product_g<G>   (a, b) staged  <-  (a, b) x (ra, rb) read from global memory, optionally stored back to global memory
                        [window multiplications, table rounds, the entry to and the exit from the Montgomery domain]

Arithmetic = csrc/kernels_basen.hpp (bn_sqr_a, bn_mul_impl, bn_b_init, bn_double, bn_add), value for value: the lane model in
tools/bn_asm/machine.py executes the generated instructions and tests/test_bn_asm.py compares them with tests/basen_model.py.

Register map (VGPRs):
    c[k]   v[2k : 2k+1]  k = 0..35     the circular window of 64-bit column accumulators
    A[k]   v[72 + k]                    this lane's block of the register operand
    N[k]   v[108 + k]                   this lane's block of M~ (the Orup multiple of n)
    v144 .. v183                        temporaries of the row loops (staged limbs, quotient digits, carries)
    v184 .. v193                        per-lane inputs, never written here (LDS addresses, lane mask, global addresses)
    U[k]   v[194 + k]                   the cross product rb * a of a base-n product: its low half, then (from global scratch) its high half
Scalar registers s30 .. s51 (see Layout)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from machine import Program, v, s, EXEC  # noqa: E402

LB = 29
MASK = (1 << LB) - 1
W = 36
BLK = 36
ROWB = BLK * 4          # bytes of one lane block in LDS
CH = 12                 # limbs per chunk of the code between the row loops


class Layout:
    """register and LDS layout for G lanes per n-sized integer"""

    def __init__(self, G):
        assert G in (2, 4)
        self.G = G
        self.bcast = "quad_perm:[0,0,2,2]" if G == 2 else "quad_perm:[0,0,0,0]"
        self.qmask32 = {2: 0x55555555, 4: 0x11111111}[G]
        self.area_b = G * ROWB              # byte offset of B() in a group's area
        # the workgroup's constant block in LDS: C3 as (value, 0) pairs, lane block by lane block (2 ROWB bytes each), then M~ (ROWB each)
        self.mt_off = G * 2 * ROWB
        self.cst_bytes = G * 3 * ROWB
        self.pair_b = G * W * 4             # bytes from the a part to the b part of a pair in global memory

    # columns, operands
    @staticmethod
    def C(col): return v(2 * (col % W), 2)
    @staticmethod
    def Clo(col): return v(2 * (col % W))
    @staticmethod
    def Chi(col): return v(2 * (col % W) + 1)
    @staticmethod
    def A(k): return v(72 + k)
    @staticmethod
    def N(k): return v(108 + k)
    # temporaries of the row loops
    @staticmethod
    def BQ(t): return v(144 + ((t // 4) % 3) * 4 + (t % 4))
    @staticmethod
    def BQquad(q): return v(144 + (q % 3) * 4, 4)
    @staticmethod
    def QD(t): return v(156 + ((t // 4) % 3) * 4 + (t % 4))
    @staticmethod
    def QDquad(q): return v(156 + (q % 3) * 4, 4)
    @staticmethod
    def XZ(i): return v(168 + 2 * (i & 1), 2)
    @staticmethod
    def X(i): return v(168 + 2 * (i & 1))
    @staticmethod
    def Z(i): return v(169 + 2 * (i & 1))
    @staticmethod
    def TL(i): return v(172 + (i & 1))
    @staticmethod
    def TQ(i): return v(174 + (i & 1))
    @staticmethod
    def SH(i): return v(176 + 2 * (i & 1), 2)
    @staticmethod
    def SHlo(i): return v(176 + 2 * (i & 1))
    @staticmethod
    def B2(i): return v(180 + (i & 1))
    VROW = v(182)        # LDS byte address of the current row of the staged operand
    VT = v(183)          # scratch address
    # per-lane inputs (the caller computes them once per kernel)
    VAREA = v(184)       # LDS byte address of the group's area (A() at +0, B() at +area_b)
    VGLO = v(185)        # gl * ROWB: this lane's block inside an area
    VGLM = v(186)        # gl ? 0xffffffff : 0: lanes that take a carry from the lane below
    VCST = v(187)        # LDS byte address of this lane's block of C3 pairs (constant block + 2 * VGLO)
    VSRC = v(188, 2)     # product: global address of this lane's block of ra (rb: + pair_b bytes)
    VDST = v(190, 2)     # product: global address of this lane's block of the destination pair
    VSCR = v(192, 2)     # product: global address of this lane's block of the group's scratch entry (the high half of the wide cross product)
    @staticmethod
    def U(k): return v(194 + k)
    LAST_VGPR = 232
    # the per-key flavour (k_enc_basen_keys: every group under a key of its own): no constant block in LDS —
    VN1 = v(187)         # n1 of this lane's key (in place of VCST)
    VKEY = v(230, 2)     # global address of this lane's block of M~ in its key's record (BnConst: C3 one integer behind)
    VMASK = v(232)       # the limb mask 2^29 - 1 (operand of v_and_b32_dpp: FUSE_AND_DPP)

    # scalar registers
    RET = s(30, 2)       # return address of the entry point
    RET2 = s(48, 2)      # return address of an internal subroutine (s32 / s33 are the compiler's stack and frame pointers)
    SINK = s(34, 2)      # carry-out of the multiply-adds (never set)
    QMASK = s(36, 2)     # lanes that write quotient digits (0: none)
    SAVE = s(38, 2)
    SROW = s(40)         # rows left
    SCNT = s(41)         # in: squarings to do
    SN1 = s(42)          # in: n1 = -n^-1 mod 2^29
    SFLAGS = s(44)       # product, in: bit 0 = the pair has a b part (P0 runs), bit 2 = the result is also stored at VDST
    LAST_SGPR = 51


def take_sqr(t, k):
    """bigint29.hpp sqr_mult: what limb k is multiplied by at sub-step position t of a squaring: 0 nothing, 1 b, 2 2b"""
    if k == t:
        return 1
    d = (k - t + W) % W
    H = W // 2
    return 2 if ((1 <= d < H) or (d == H and t < H)) else 0


# (cross-lane move, limb mask) as ONE v_and_b32_dpp in the row loops of the products on M~: 2 of the 8 bookkeeping instructions of a
# sub-step.  The compiled kernels tried this in round 4 (ZKP_AND_DPP) and gained nothing at the power cap; A/B on the engine: profiles/r06/.
FUSE_AND_DPP = os.environ.get("ZKP_BN_ASM_FUSE", "1") != "0"


class Gen:
    def __init__(self, G, fuse=None):
        self.L = Layout(G)
        self.G = G
        self.p = Program()
        self.uid = 0
        self.fuse = FUSE_AND_DPP if fuse is None else fuse

    def lbl(self, name):
        return f".Lzkp_bn{self.G}_{name}"

    def fresh_label(self, name):
        self.uid += 1
        return self.lbl(f"{name}{self.uid}")

    # ------------------------------------------------------------------------------------------------ the row loops
    def a_products(self, kind, t):
        """the A-half multiply-adds of sub-step position t: list of (column, A register index, multiplier kind 1 = b, 2 = 2b)"""
        out = []
        t %= W
        for k in range(W):
            m = 1 if kind == "mul" else take_sqr(t, k)
            if m:
                out.append(((t + k) % W, k, m))
        return out

    def rows(self, kind):
        """One n-sized product on M~, G rows of W sub-steps: c += A * (staged operand) + M~ * (quotient digits), one column finished per
        sub-step (bigint29.hpp montmul / montsqr, kernels_basen.hpp bn_mul_impl / bn_sqr_a).

        Entries:  <kind>_rows_zero  columns start at 0      mul_rows_init  columns start at the values in c[]
        In: A[], N[], VROW = LDS byte address of row 0 of the staged operand, QMASK.  Out: c[k] = the columns the carry chain reads in
        order.  Returns through RET2.

        The loop is software-pipelined by half a sub-step: block t issues M~ * q_(t-1), finishes column t-1 and issues A * b_t; the
        dependent chain  digit -> N[0] q -> shift -> add -> mask -> broadcast  is spread over the block's multiply-adds."""
        L, p = self.L, self.p
        mul = kind == "mul"

        def mad(col, a, b, fresh):
            """c[col] += a * b; a column listed in `fresh` takes its first value from there (a register pair or the constant 0)"""
            src = fresh.pop(col % W, None)
            p.v_mad_u64_u32(L.C(col), L.SINK, a, b, L.C(col) if src is None else src)

        def bmul(t, m):
            return L.BQ(t) if m == 1 else L.B2(t)

        def read_quad(q, row_off=0):
            p.ds_read(128, L.BQquad(q), L.VROW, row_off + (q % 9) * 16)

        def prologue(zero):
            # position 0 of row 0: nothing to finish yet
            read_quad(0)
            read_quad(1)
            p.s_mov_b32(L.SROW, self.G)
            p.s_waitcnt(lgkm=1)
            fresh = {k: 0 for k in range(W)} if zero else {}
            if not mul:
                p.v_lshlrev_b32(L.B2(0), 1, L.BQ(0))
            prods = self.a_products(kind, 0)
            prods.sort(key=lambda x: (x[0] != 0, x[0]))           # the product into column 0 first: it decides the first digit
            assert prods[0][0] == 0
            for n_, (col, k, m) in enumerate(prods):
                mad(col, L.A(k), bmul(0, m), fresh)
                if n_ == 3 and not self.fuse:
                    p.v_and_b32(L.TQ(0), MASK, L.Clo(0))
            for col in sorted(fresh):                             # columns a squaring does not touch at position 0
                p.v_mov_b64(L.C(col), 0)
            if self.fuse:
                p.v_and_b32_dpp(L.QD(0), L.Clo(0), L.VMASK, L.bcast)
            else:
                p.v_mov_b32_dpp(L.QD(0), L.TQ(0), L.bcast)
            if not mul:
                p.v_lshlrev_b32(L.B2(1), 1, L.BQ(1))
            for (col, k, m) in self.a_products(kind, 1):          # the product of position 1 that lands in column 1: block 1's digit waits for it
                if col == 1:
                    p.v_mad_u64_u32(L.C(1), L.SINK, L.A(k), bmul(1, m), L.C(1))
            p.s_branch(self.lbl(f"{kind}_loop"))

        def block(t, carried, tail=False, settle=False):
            """finish column t-1 with digit q_(t-1); issue the A half of position t (not in the tail block behind the last row).
            carried: a fresh top column whose first multiply-add comes in this block; settle: leave none behind (the loop's back edge)"""
            tm = t - 1
            qprev = L.QD(tm % W)
            fresh = dict(carried)
            fill = []
            lead = [lambda: mad(tm, L.N(0), qprev, fresh), lambda: mad(tm + 1, L.N(1), qprev, fresh)]
            for k in range(2, W):
                fill.append(lambda k=k: mad(tm + k, L.N(k), qprev, fresh))
            pre_next = []
            fresh_mad = None
            if not tail:
                for (col, k, m) in self.a_products(kind, t):
                    if col == t % W:
                        continue                                   # issued at the end of the block before (it decides this block's digit)
                    th = (lambda col=col, k=k, m=m: mad(col, L.A(k), bmul(t, m), fresh))
                    if col == tm % W:
                        fresh_mad = th                             # the first value of the fresh top column
                    else:
                        fill.append(th)
                # what the NEXT block's digit waits for is issued at the end of this one — except across a row's end: behind the last
                # row there is no next position, so block W issues its own (right behind its two leading multiply-adds)
                if t + 1 != W:
                    for (col, k, m) in self.a_products(kind, t + 1):
                        if col == (t + 1) % W:
                            pre_next.append(lambda col=col, k=k, m=m: mad(col, L.A(k), bmul(t + 1, m), fresh))
                if t == W:
                    for (col, k, m) in self.a_products(kind, t):
                        if col == t % W:
                            lead.append(lambda col=col, k=k, m=m: mad(col, L.A(k), bmul(t, m), fresh))
            i2 = t & 1

            def finish_mask(): p.v_and_b32(L.TL(i2), MASK, L.Clo(tm))
            def finish_shift(): p.v_lshrrev_b64(L.SH(i2), LB, L.C(tm))
            def carry(): p.v_lshl_add_u64(L.C(tm + 1), L.C(tm + 1), 0, L.SH(i2))

            def pass_down():
                direct = tail or (settle and not fresh_mad)
                dst = L.Clo(tm) if direct else L.X(i2)
                if self.fuse:
                    p.v_and_b32_dpp(dst, L.Clo(tm), L.VMASK, "row_shl:1")
                else:
                    p.v_mov_b32_dpp(dst, L.TL(i2), "row_shl:1")
                if direct:
                    p.v_mov_b32(L.Chi(tm), 0)
                else:
                    fresh[tm % W] = L.XZ(i2)

            def digit_mask(): p.v_and_b32(L.TQ(i2), MASK, L.Clo(t))

            def digit():
                if self.fuse:
                    p.v_and_b32_dpp(L.QD(t % W), L.Clo(t), L.VMASK, L.bcast)
                else:
                    p.v_mov_b32_dpp(L.QD(t % W), L.TQ(i2), L.bcast)

            if not tail and not mul and t == W:
                p.v_lshlrev_b32(L.B2(t), 1, L.BQ(t))
            if self.fuse:
                seq = lead + ["f", "f", finish_shift, "f", "f", "f", carry, pass_down, "f", "f", "f"]
                if not tail:
                    seq += [fresh_mad if fresh_mad else "f", "f", digit]
            else:
                seq = lead + ["f", "f", finish_mask, finish_shift, "f", "f", "f", carry, pass_down, "f", "f", "f"]
                if not tail:
                    seq += [digit_mask, fresh_mad if fresh_mad else "f", "f", "f", digit]
            tq = t % W
            if not tail and tq % 4 == 0:
                nxt = tq // 4 + 1                                  # the quad after this one; behind a row's ninth comes the next row's first
                seq.insert(len(lead), (lambda: read_quad(nxt % 9, ROWB if nxt == 9 else 0)))
            it = iter(fill)
            for x in seq:
                if x == "f":
                    th = next(it, None)
                    if th:
                        th()
                    else:
                        p.s_nop(0)
                else:
                    x()
            for th in it:
                th()
            if not tail:
                if tq % 4 == 3:
                    p.s_waitcnt(lgkm=0)                            # the staged limbs of the next quad
                if not mul and t + 1 != W:
                    p.v_lshlrev_b32(L.B2(t + 1), 1, L.BQ(t + 1))
                for th in pre_next:
                    th()
                if tq % 4 == 3:
                    # the digits of a quad of sub-steps go out once the last one exists (lane 0 of every group; QMASK = 0: nobody)
                    p.s_and_saveexec_b64(L.SAVE, L.QMASK)
                    p.ds_write(128, L.VROW, L.QDquad(tq // 4), (tq - 3) * 4)
                    p.s_mov_b64(EXEC, L.SAVE)
            return fresh

        p.label(self.lbl(f"{kind}_rows_zero"))
        prologue(True)
        if mul:
            p.label(self.lbl(f"{kind}_rows_init"))
            prologue(False)
        # ---- the loop: blocks 1 .. 35 of a row, then block 0 of the next row (or the tail behind the last row)
        p.label(self.lbl(f"{kind}_loop"))
        carried = {}
        for t in range(1, W):
            carried = block(t, carried)
        assert not carried, "position 35 always multiplies into the fresh column"
        p.s_sub_u32(L.SROW, L.SROW, 1)
        p.s_cmp_eq_u32(L.SROW, 0)
        p.s_cbranch_scc1(self.lbl(f"{kind}_tail"))
        p.v_add_u32(L.VROW, ROWB, L.VROW)
        left = block(W, {}, settle=True)
        assert not left
        p.s_branch(self.lbl(f"{kind}_loop"))
        p.label(self.lbl(f"{kind}_tail"))
        block(W, {}, tail=True)
        p.s_setpc_b64(L.RET2)


    def rows_wide(self):
        """The cross product rb * a of a base-n product as the EXACT integer HI R' + LO — no reduction, half the multiply-adds of a product on
        M~ (kernels_basen.hpp bn_mul_full; profiles/r05/experiments/README.md: the variant the compiler could not hold in registers).
        Every sub-step finishes one digit of LO in lane 0's bottom column; the lane whose block that digit belongs to (row s = its gl) keeps
        it.  LO joins the b side's initial columns before that side's ONE reduction, HI is added to its result:
            (ra b + LO - Q n1 + q M~) / R' + HI  ==  (ra b + rb a - Q n1) / R'   (mod n).
        In: A[] = rb, VROW = row 0 of the staged a.  Out: U[] = this lane's block of LO, c[] = the columns of HI.  Returns through RET2."""
        L, p = self.L, self.p
        from machine import VCC
        VDOWN = L.VT                                                # (the scratch address register is free inside the row loop)

        def mad(col, a, b, fresh):
            src = fresh.pop(col % W, None)
            p.v_mad_u64_u32(L.C(col), L.SINK, a, b, L.C(col) if src is None else src)

        def read_quad(q, row_off=0):
            p.ds_read(128, L.BQquad(q), L.VROW, row_off + (q % 9) * 16)

        def block(t, tail=False):
            tm = t - 1
            i2 = t & 1
            fresh = {}
            fill = [] if tail else [(lambda k=k: mad(t + k, L.A(k), L.BQ(t), fresh)) for k in range(W - 1)]
            top = None if tail else (lambda: mad(t + W - 1, L.A(W - 1), L.BQ(t), fresh))

            def finish_mask(): p.v_and_b32(L.TL(i2), MASK, L.Clo(tm))
            # what goes down to the lane below: nothing from lane 0 of a group — the top lane of the group below opens its fresh column at
            # 0 (nothing was reduced: a group's bottom limb is not zero here as it is in a product on M~)
            def down_mask(): p.v_and_b32(L.B2(i2), VDOWN, L.Clo(tm))
            def finish_shift(): p.v_lshrrev_b64(L.SH(i2), LB, L.C(tm))
            def carry(): p.v_lshl_add_u64(L.C(tm + 1), L.C(tm + 1), 0, L.SH(i2))
            def bcast(): p.v_mov_b32_dpp(L.TQ(i2), L.TL(i2), L.bcast)
            def capture(): p.v_cndmask_b32(L.U(tm % W), L.U(tm % W), L.TQ(i2))

            def pass_down():
                if tail:
                    p.v_mov_b32_dpp(L.Clo(tm), L.B2(i2), "row_shl:1")
                    p.v_mov_b32(L.Chi(tm), 0)
                else:
                    p.v_mov_b32_dpp(L.X(i2), L.B2(i2), "row_shl:1")
                    fresh[tm % W] = L.XZ(i2)

            seq = ["f", "f", "f", finish_mask, down_mask, finish_shift, "f", "f", carry, bcast, "f", "f", pass_down, capture, "f", "f"]
            if top:
                seq.append(top)
            tq = t % W
            if not tail and tq % 4 == 0:
                nxt = tq // 4 + 1
                seq.insert(0, (lambda: read_quad(nxt % 9, ROWB if nxt == 9 else 0)))
            it = iter(fill)
            for x in seq:
                if x == "f":
                    th = next(it, None)
                    if th:
                        th()
                    else:
                        p.s_nop(0)
                else:
                    x()
            for th in it:
                th()
            if not tail and tq % 4 == 3:
                p.s_waitcnt(lgkm=0)
            assert not fresh

        p.label(self.lbl("wide_rows"))
        read_quad(0)
        read_quad(1)
        p.s_mov_b32(L.SROW, self.G)
        p.s_mov_b32(("vcc_lo", 0, 1), L.qmask32)                   # the lanes whose block row 0 is: lane 0 of every group
        p.s_mov_b32(("vcc_hi", 0, 1), L.qmask32)
        p.v_and_b32(VDOWN, MASK, L.VGLM)                            # the limb mask, or 0 in lane 0 of a group
        p.s_waitcnt(lgkm=1)
        fresh = {k: 0 for k in range(W)}
        for k in range(W):
            mad(k, L.A(k), L.BQ(0), fresh)
        p.label(self.lbl("wide_loop"))
        for t in range(1, W):
            block(t)
        p.s_sub_u32(L.SROW, L.SROW, 1)
        p.s_cmp_eq_u32(L.SROW, 0)
        p.s_cbranch_scc1(self.lbl("wide_tail"))
        p.v_add_u32(L.VROW, ROWB, L.VROW)
        block(W)
        p.s_lshl_b64(VCC, VCC, 1)                                   # the next row's digits belong to the next lane of every group
        p.s_branch(self.lbl("wide_loop"))
        p.label(self.lbl("wide_tail"))
        block(W, tail=True)
        p.s_setpc_b64(L.RET2)

    # ------------------------------------------------------------------------------------------------ between the row loops
    def chain_chunk(self, base, dst0, state):
        """carry chain over columns base .. base + CH - 1: limb k -> v[dst0 + k - base]; the carry travels in SH(0)"""
        L, p = self.L, self.p
        cy = L.SH(0)
        for k in range(base, base + CH):
            if state["first"]:
                state["first"] = False
            else:
                p.v_lshl_add_u64(L.C(k), L.C(k), 0, cy)
            p.v_and_b32(v(dst0 + k - base), MASK, L.Clo(k))
            p.v_lshrrev_b64(cy, LB, L.C(k))

    def carry_to_next_lane(self, dst):
        """the carry out of this lane's block (SH(0), tiny) -> dst of the next lane of the group; lane 0 of a group receives 0"""
        L, p = self.L, self.p
        p.s_nop(1)
        p.v_mov_b32_dpp(dst, L.SHlo(0), "row_shr:1")
        p.v_and_b32(dst, dst, L.VGLM)

    def fin_a_init_b(self):
        """The a side of a squaring or product is done (columns in c[]): its result a' goes into A() of the group IN PLACE of the quotient
        digits that wait there, and the b side's initial columns c_k = C3_k + (2^29 - Q_k) n1 take the columns' place — twelve limbs
        at a time, so that neither the digits nor a' ever exist as a 36-register array (kernels_basen.hpp: the carry chain of
        bn_sqr_a / bn_mul_impl, bn_b_init, bn_stage of `pend`).  The carry out of a lane's block lands on limb 0 of the next lane's
        block with one LDS add.  Returns through RET2."""
        L, p = self.L, self.p
        p.label(self.lbl("fin_a_init_b"))
        p.v_add_u32(L.VT, L.VAREA, L.VGLO)                         # this lane's block of A()
        Q0, R0 = 144, 156                                          # the digits of a chunk: v144 .. v155; its limbs of a': v156 .. v167
        state = {"first": True}
        for base in range(0, W, CH):
            for j in range(0, CH, 4):
                p.ds_read(128, v(Q0 + j, 4), L.VT, (base + j) * 4)
            self.chain_chunk(base, R0, state)
            for j in range(0, CH, 4):
                p.ds_write(128, L.VT, v(R0 + j, 4), (base + j) * 4)
            for k in range(base, base + CH, 2):                    # C3 pairs straight into the columns the chain has left
                p.ds_read(128, v(2 * k, 4), L.VCST, k * 8)
            p.s_waitcnt(lgkm=CH // 2 + CH // 4)                    # the digits are here (the writes and the C3 reads are behind them)
            for k in range(base, base + CH):
                p.v_sub_u32(v(Q0 + k - base), 1 << LB, v(Q0 + k - base))
            p.s_waitcnt(lgkm=0)
            nolo = self.fresh_label("fin_a_nolo")
            p.s_bitcmp1_b32(L.SFLAGS, 0)
            p.s_cbranch_scc0(nolo)
            for k in range(base, base + CH):                       # + LO of the wide cross product (limbs below 2^29 on limbs below 2^29)
                p.v_add_u32(L.Clo(k), L.Clo(k), L.U(k))
            p.label(nolo)
            for k in range(base, base + CH):
                p.v_mad_u64_u32(L.C(k), L.SINK, v(Q0 + k - base), L.SN1, L.C(k))
        self.carry_to_next_lane(L.TL(1))
        p.ds_add_u32(L.VT, L.TL(1), 0)
        p.s_setpc_b64(L.RET2)

    def fin_b(self):
        """The b side is done: carry chain of c[] (+ the parked cross product U when SFLAGS bit 0 is set) -> B() of the group, twelve limbs at
        a time; with SFLAGS bit 2 also to global memory at VDST + pair_b (the b part of a table entry / of the raw pair).
        Returns through RET2."""
        L, p = self.L, self.p
        p.label(self.lbl("fin_b"))
        p.v_add_u32(L.VT, L.VAREA, L.VGLO)
        p.s_bitcmp1_b32(L.SFLAGS, 0)
        p.s_cbranch_scc0(self.lbl("fin_b_chain"))
        p.s_waitcnt(vm=0)                                           # (the high half of the cross product, on its way back from the scratch entry)
        for k in range(W):                                          # c_k += U_k
            p.v_mad_u64_u32(L.C(k), L.SINK, L.U(k), 1, L.C(k))
        p.label(self.lbl("fin_b_chain"))
        R0 = 156
        state = {"first": True}
        for base in range(0, W, CH):
            self.chain_chunk(base, R0, state)
            for j in range(0, CH, 4):
                p.ds_write(128, L.VT, v(R0 + j, 4), L.area_b + (base + j) * 4)
            skip = self.fresh_label("fin_b_nostore")
            p.s_bitcmp1_b32(L.SFLAGS, 2)
            p.s_cbranch_scc0(skip)
            for j in range(0, CH, 4):
                p.global_store(4, L.VDST, v(R0 + j, 4), L.pair_b + (base + j) * 4)
            p.label(skip)
            p.s_nop(1)                                              # (the stores read their data before the next chunk overwrites it)
        self.carry_to_next_lane(L.TL(1))
        p.ds_add_u32(L.VT, L.TL(1), L.area_b)
        p.s_bitcmp1_b32(L.SFLAGS, 2)
        p.s_cbranch_scc0(self.lbl("fin_b_done"))
        p.ds_read(32, L.TL(0), L.VT, L.area_b)                      # limb 0 with the carry in place
        p.s_waitcnt(lgkm=0)
        p.global_store(1, L.VDST, L.TL(0), L.pair_b)
        p.label(self.lbl("fin_b_done"))
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_setpc_b64(L.RET2)

    def double_a(self):
        """A[] <- 2 A[], limbs normalised again (kernels_basen.hpp bn_double) — WITHOUT a carry chain: 2 a_k is even, so
        (2 a_k mod 2^29) + carry_in < 2^29 whatever the carry: a carry never ripples, and the carry out of limb k is just the top bit(s) of
        a_k: R_k = ((a_k << 1) & MASK) + (a_(k-1) >> 28).  Three independent instructions per limb instead of four dependent ones (a lone
        wavefront in a dependent chain leaves its SIMD half idle).  A limb that arrives almost normalised (limb 0 of a lane's block after
        the carry of the lane below: < 2^29 + 2^8) hands up 2 or 3 instead of 0 or 1: the result stays within the same tolerance."""
        L, p = self.L, self.p
        p.label(self.lbl("double_a"))
        t = [L.TL(0), L.TL(1), L.TQ(0), L.TQ(1)]
        # the carry into limb 0 comes from the top limb of the lane below (before that limb is overwritten)
        p.v_lshrrev_b32(L.SHlo(0), LB - 1, L.A(W - 1))
        for k in range(W - 1, 0, -1):                               # top down: limb k reads limb k-1 before it is rewritten
            tk = t[k & 3]
            p.v_lshrrev_b32(tk, LB - 1, L.A(k - 1))
            p.v_and_b32(L.A(k), MASK >> 1, L.A(k))
            p.v_lshl_add_u32(L.A(k), L.A(k), 1, tk)
        p.v_mov_b32_dpp(L.TL(0), L.SHlo(0), "row_shr:1")
        p.v_and_b32(L.A(0), MASK >> 1, L.A(0))
        p.v_and_b32(L.TL(0), L.TL(0), L.VGLM)
        p.v_lshl_add_u32(L.A(0), L.A(0), 1, L.TL(0))
        p.s_setpc_b64(L.RET2)

    def lane_setup(self, per_key=False):
        """M~ into N[] from the constant block (per-key flavour: from the key's record in global memory); the zero halves of the
        fresh-column pairs"""
        L, p = self.L, self.p
        if per_key:
            for j in range(0, W, 4):
                p.global_load(4, v(108 + j, 4), L.VKEY, j * 4)
        else:
            p.v_sub_u32(L.VT, L.VCST, L.VGLO)                       # constant block + this lane's ROWB
            for j in range(0, W, 4):
                p.ds_read(128, v(108 + j, 4), L.VT, L.mt_off + j * 4)
        p.v_mov_b32(L.Z(0), 0)
        p.v_mov_b32(L.Z(1), 0)
        p.v_mov_b32(L.VMASK, MASK)

    def load_c3(self, first_reg):
        """per-key flavour: this lane's block of its key's C3 -> 36 registers from `first_reg`"""
        L, p = self.L, self.p
        for j in range(0, W, 4):
            p.global_load(4, v(first_reg + j, 4), L.VKEY, L.pair_b + j * 4)

    def fin_a_init_b_k(self):
        """fin_a_init_b of the per-key flavour: the b side's additive block — C3 of the lane's key, plus LO of the wide cross product in a
        product — waits in U[]; c_k = (2^29 - Q_k) n1 + U_k with n1 in a vector register."""
        L, p = self.L, self.p
        p.label(self.lbl("fin_a_init_b_k"))
        p.v_add_u32(L.VT, L.VAREA, L.VGLO)
        p.s_waitcnt(vm=0)                                           # (C3 on its way into U[])
        Q0, R0 = 144, 156
        state = {"first": True}
        for base in range(0, W, CH):
            for j in range(0, CH, 4):
                p.ds_read(128, v(Q0 + j, 4), L.VT, (base + j) * 4)
            self.chain_chunk(base, R0, state)
            for j in range(0, CH, 4):
                p.ds_write(128, L.VT, v(R0 + j, 4), (base + j) * 4)
            p.s_waitcnt(lgkm=CH // 4)
            for k in range(base, base + CH):
                p.v_sub_u32(v(Q0 + k - base), 1 << LB, v(Q0 + k - base))
            for k in range(base, base + CH):
                p.v_mad_u64_u32(L.C(k), L.SINK, v(Q0 + k - base), L.VN1, 0)
            for k in range(base, base + CH):
                p.v_mad_u64_u32(L.C(k), L.SINK, L.U(k), 1, L.C(k))
        self.carry_to_next_lane(L.TL(1))
        p.ds_add_u32(L.VT, L.TL(1), 0)
        p.s_setpc_b64(L.RET2)

    def set_qmask(self, on):
        L, p = self.L, self.p
        if on:
            p.s_mov_b32(s(36), L.qmask32)
            p.s_mov_b32(s(37), L.qmask32)
        else:
            p.s_mov_b64(L.QMASK, 0)

    # ------------------------------------------------------------------------------------------------ entry points
    def sqr_run(self, per_key=False):
        """(a, b) staged <- (a, b)^(2^SCNT).  In: v184 .. v187 (Layout), s41 = count (>= 1), s42 = n1 (per-key flavour: v187 = n1,
        v[230:231] = the key's record)."""
        L, p = self.L, self.p
        k = "_k" if per_key else ""
        p.label(f"zkp_bn_sqr_run{k}_g{self.G}")
        p.s_waitcnt(vm=0, lgkm=0)                                   # (what the caller left in flight: its stores feed this code's loads)
        self.lane_setup(per_key)
        p.s_mov_b32(L.SFLAGS, 0)
        p.label(self.lbl("sqr_next" + k))
        if per_key:
            self.load_c3(194)                                       # C3 -> U[]: it has the whole a side to arrive
        p.v_add_u32(L.VT, L.VAREA, L.VGLO)
        for j in range(0, W, 4):
            p.ds_read(128, v(72 + j, 4), L.VT, j * 4)
        p.v_mov_b32(L.VROW, L.VAREA)
        self.set_qmask(True)
        if per_key:
            p.s_waitcnt(vm=W // 4, lgkm=0)                          # (M~ is here — the first pass — ; C3 may still be under way)
        else:
            p.s_waitcnt(lgkm=0)
        p.s_call_b64(L.RET2, self.lbl("sqr_rows_zero"))          # a^2: digits over the staged a
        p.s_call_b64(L.RET2, self.lbl("double_a"))               # 2 a: the register operand of the b side
        p.s_call_b64(L.RET2, self.lbl("fin_a_init_b" + k))       # a' staged; columns = C3 + (2^29 - Q) n1
        p.v_add_u32(L.VROW, L.area_b, L.VAREA)
        self.set_qmask(False)
        p.s_call_b64(L.RET2, self.lbl("mul_rows_init"))          # 2 a b + the quotient term
        p.s_call_b64(L.RET2, self.lbl("fin_b"))                  # b' staged
        p.s_sub_u32(L.SCNT, L.SCNT, 1)
        p.s_cmp_lg_u32(L.SCNT, 0)
        p.s_cbranch_scc1(self.lbl("sqr_next" + k))
        p.s_setpc_b64(L.RET)

    def product(self, per_key=False):
        """(a, b) staged <- (a, b) x (ra, rb), the pair read from global memory at VSRC.  In: v184 .. v193, s42 = n1, s44 = flags:
        bit 0 = the pair has a b part, bit 2 = the result is also stored to global memory at VDST.  Slots as in kernels_basen.hpp
        (k_enc_basen): P0 rb * a -> U (registers), P1 ra * a with the digits out, P2 ra * b with the quotient term, + U."""
        L, p = self.L, self.p
        k = "_k" if per_key else ""
        p.label(f"zkp_bn_product{k}_g{self.G}")
        p.s_waitcnt(vm=0, lgkm=0)
        self.lane_setup(per_key)
        p.s_bitcmp1_b32(L.SFLAGS, 0)
        p.s_cbranch_scc0(self.lbl("prod_p1" + k))
        # ---- P0: rb * a = HI R' + LO, exactly: LO -> U[], HI -> the group's scratch entry (it comes back into U[] once LO is spent)
        for j in range(0, W, 4):
            p.global_load(4, v(72 + j, 4), L.VSRC, L.pair_b + j * 4)
        p.v_mov_b32(L.VROW, L.VAREA)
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_call_b64(L.RET2, self.lbl("wide_rows"))
        R0 = 156
        state = {"first": True}
        for base in range(0, W, CH):
            self.chain_chunk(base, R0, state)
            if base == 0:
                p.v_mov_b32(L.TQ(0), v(R0))                         # limb 0 waits for the carry of the lane below
            for j in range(0, CH, 4):
                p.global_store(4, L.VSCR, v(R0 + j, 4), (base + j) * 4)
            p.s_nop(1)
        self.carry_to_next_lane(L.TL(1))
        p.v_add_u32(L.TQ(0), L.TQ(0), L.TL(1))
        p.global_store(1, L.VSCR, L.TQ(0), 0)
        if per_key:
            # the b side's additive block: LO + C3 of the lane's key (the columns are free between the row loops: C3 lands there)
            self.load_c3(0)
            p.s_waitcnt(vm=0)
            for i in range(W):
                p.v_add_u32(L.U(i), L.U(i), v(i))
            p.s_branch(self.lbl("prod_p1_go" + k))
        # ---- P1: ra * a, digits out
        p.label(self.lbl("prod_p1" + k))
        if per_key:
            self.load_c3(194)                                       # no cross product: the additive block is C3 alone
            p.label(self.lbl("prod_p1_go" + k))
        for j in range(0, W, 4):
            p.global_load(4, v(72 + j, 4), L.VSRC, j * 4)
        p.v_mov_b32(L.VROW, L.VAREA)
        self.set_qmask(True)
        p.s_waitcnt(vm=0, lgkm=0)
        p.s_call_b64(L.RET2, self.lbl("mul_rows_zero"))
        p.s_call_b64(L.RET2, self.lbl("fin_a_init_b" + k))
        p.s_bitcmp1_b32(L.SFLAGS, 2)
        p.s_cbranch_scc0(self.lbl("prod_p2" + k))
        # the a part as staged (the carry into limb 0 is in place: LDS operations of a wavefront execute in order) -> global memory
        p.v_add_u32(L.VT, L.VAREA, L.VGLO)
        for base in range(0, W, CH):
            for j in range(0, CH, 4):
                p.ds_read(128, v(144 + j, 4), L.VT, (base + j) * 4)
            p.s_waitcnt(lgkm=0)
            for j in range(0, CH, 4):
                p.global_store(4, L.VDST, v(144 + j, 4), (base + j) * 4)
            p.s_nop(1)
        # ---- P2: ra * b + the quotient term + LO (in the columns already), + HI at the end
        p.label(self.lbl("prod_p2" + k))
        p.s_bitcmp1_b32(L.SFLAGS, 0)
        p.s_cbranch_scc0(self.lbl("prod_p2_rows" + k))
        p.s_waitcnt(vm=0)                                           # (the stores of HI: same wavefront, same addresses)
        for j in range(0, W, 4):
            p.global_load(4, v(194 + j, 4), L.VSCR, j * 4)
        p.label(self.lbl("prod_p2_rows" + k))
        p.v_add_u32(L.VROW, L.area_b, L.VAREA)
        self.set_qmask(False)
        p.s_call_b64(L.RET2, self.lbl("mul_rows_init"))
        p.s_call_b64(L.RET2, self.lbl("fin_b"))
        p.s_setpc_b64(L.RET)

    def build(self, per_key=True):
        """per_key=False: the shared-key entry points only (what k_enc_basen executes: the size that has to fit the instruction cache).
        Layout: [shared-key entry points and their fin_a_init_b] [the row loops, fin_b, double_a: both flavours] [per-key entry points and
        theirs] — what either kernel executes is one contiguous stretch of the text."""
        self.sqr_run()
        self.product()
        self.fin_a_init_b()
        self.rows("sqr")
        self.rows("mul")
        self.rows_wide()
        self.fin_b()
        self.double_a()
        if per_key:
            self.fin_a_init_b_k()
            self.sqr_run(per_key=True)
            self.product(per_key=True)
        return self.p


HEADER = """// GENERATED by tools/bn_asm/gen.py — do not edit; `python tools/bn_asm/gen.py` rewrites it, tests/test_bn_asm.py holds it to the generator.
// The fixed-register base-n product engine of k_enc_basen<{G}> (gfx950, 36 limbs per lane): see the generator for the register map, the
// arithmetic (csrc/kernels_basen.hpp, value for value) and the lane model that executes these very instructions on the CPU.
"""


def inc_text(G):
    prog = Gen(G).build()
    out = [HEADER.format(G=G), f"// {prog.count()} instructions, about {prog.bytes_estimate() // 1024} KB of code\n"]
    for ln in prog.lines():
        out.append('"' + ln + '\\n"\n')
    return prog, "".join(out)


def clobber_text(G):
    """the registers the engine writes, as the clobber list of the C++ call sites (the per-lane inputs v184 .. v191 and the scalar inputs
    s41 / s42 / s44 are operands there, not clobbers)"""
    L = Layout(G)
    vs = [f'"v{i}"' for i in range(0, 184)] + [f'"v{i}"' for i in range(194, L.LAST_VGPR + 1) if i not in (230, 231)]      # (v230 / v231: an input of the per-key flavour)
    ss = [f'"s{i}"' for i in range(30, L.LAST_SGPR + 1) if i not in (32, 33, 41, 42, 44)]
    out = ["// GENERATED by tools/bn_asm/gen.py: what the engine of kernels_basen_asm_g%d.inc writes\n" % G]
    regs = vs + ss + ['"vcc"', '"scc"', '"memory"']
    for i in range(0, len(regs), 16):
        out.append(", ".join(regs[i:i + 16]) + (",\n" if i + 16 < len(regs) else "\n"))
    return "".join(out)


def inc_path(G):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return os.path.join(root, "zk-paillier_amd", "csrc", f"kernels_basen_asm_g{G}.inc")


if __name__ == "__main__":
    for G in (2, 4):
        prog, text = inc_text(G)
        with open(inc_path(G), "w") as f:
            f.write(text)
        with open(inc_path(G).replace(".inc", "_clobbers.inc"), "w") as f:
            f.write(clobber_text(G))
        print(inc_path(G), prog.count(), "instructions,", prog.bytes_estimate(), "bytes")
