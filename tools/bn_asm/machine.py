"""A tiny gfx950 assembler-with-simulator for the fixed-register base-n product engine (tools/bn_asm/gen.py).

A Program is a list of instructions written through one method per mnemonic; it can be printed as assembler text (the
committed csrc/kernels_basen_asm_g*.inc) and EXECUTED on a 64-lane value model of one wavefront: vector registers, the
scalar registers the engine uses, exec, scc, LDS and global memory as word arrays.  The model ignores time; what it does
check beside the arithmetic is what the hardware will not check for hand-written code:

  * every v_mad_u64_u32 sum stays below 2^64 (the column-capacity argument of csrc/bigint29.hpp),
  * the manually inserted wait states of the gfx940 ISA guide that apply here (a VALU write of a VGPR followed by a DPP
    read of it needs two wait states; a VALU write of data registers of a pending DS write of more than 64 bits),
  * every LDS / global load is waited for (s_waitcnt) before its destination is read or overwritten,
  * 64-bit register operands are even-aligned.

Operands: v(i) one VGPR, v(i, n) a tuple of n; s(i) / s(i, n) the same for SGPRs; plain ints are inline constants or
literals; strings are labels."""

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def v(i, n=1):
    assert 0 <= i and i + n <= 256, (i, n)
    return ("v", i, n)


def s(i, n=1):
    assert 0 <= i and i + n <= 102, (i, n)
    return ("s", i, n)


EXEC = ("exec", 0, 2)
VCC = ("vcc", 0, 2)
OFF = ("off", 0, 0)


def is_v(o):
    return isinstance(o, tuple) and o[0] == "v"


def is_s(o):
    return isinstance(o, tuple) and o[0] == "s"


def fmt(o):
    if isinstance(o, int):
        return str(o) if -16 <= o <= 64 else hex(o & M32)
    if isinstance(o, str):
        return o
    k, i, n = o
    if k in ("exec", "vcc", "off", "vcc_lo", "vcc_hi"):
        return k
    return f"{k}{i}" if n == 1 else f"{k}[{i}:{i + n - 1}]"


class Ins:
    __slots__ = ("op", "ops", "mods", "comment", "reads", "writes", "kind")

    def __init__(self, op, ops, mods="", reads=(), writes=(), kind="valu", comment=""):
        self.op, self.ops, self.mods, self.comment = op, ops, mods, comment
        self.reads, self.writes, self.kind = reads, writes, kind

    def text(self):
        t = self.op
        if self.ops:
            t += " " + ", ".join(fmt(o) for o in self.ops)
        if self.mods:
            t += " " + self.mods
        return t


def regs_of(o):
    """the VGPR / SGPR indices an operand names"""
    if isinstance(o, tuple) and o[0] in ("v", "s"):
        return [(o[0], o[1] + j) for j in range(o[2])]
    return []


class Program:
    def __init__(self):
        self.ins = []
        self.labels = {}

    # ---------------------------------------------------------------- emission
    def _e(self, op, ops, mods="", rd=(), wr=(), kind="valu", comment=""):
        reads = [r for o in rd for r in regs_of(o)]
        writes = [r for o in wr for r in regs_of(o)]
        for o in list(rd) + list(wr):
            if isinstance(o, tuple) and o[0] == "v" and o[2] == 2:
                assert o[1] % 2 == 0, f"{op}: 64-bit VGPR operand {fmt(o)} is not even-aligned"
        self.ins.append(Ins(op, ops, mods, reads, writes, kind, comment))

    def label(self, name):
        assert name not in self.labels, name
        self.labels[name] = len(self.ins)
        self.ins.append(Ins(name + ":", (), kind="label"))

    def comment(self, text):
        self.ins.append(Ins("; " + text, (), kind="comment"))

    # VALU
    def v_mad_u64_u32(self, d, sd, a, b, c):
        self._e("v_mad_u64_u32", (d, sd, a, b, c), rd=(a, b, c), wr=(d, sd))

    def _vop2(self, op, d, a, b):
        self._e(op, (d, a, b), rd=(a, b), wr=(d,))

    def v_and_b32(self, d, a, b): self._vop2("v_and_b32_e32", d, a, b)
    def v_add_u32(self, d, a, b): self._vop2("v_add_u32_e32", d, a, b)
    def v_sub_u32(self, d, a, b): self._vop2("v_sub_u32_e32", d, a, b)          # a - b
    def v_lshlrev_b32(self, d, sh, a): self._vop2("v_lshlrev_b32_e32", d, sh, a)
    def v_lshrrev_b32(self, d, sh, a): self._vop2("v_lshrrev_b32_e32", d, sh, a)
    def v_mul_u32_u24(self, d, a, b): self._vop2("v_mul_u32_u24_e32", d, a, b)
    def v_mov_b32(self, d, a): self._e("v_mov_b32_e32", (d, a), rd=(a,), wr=(d,))
    def v_mov_b64(self, d, a): self._e("v_mov_b64_e32", (d, a), rd=(a,), wr=(d,))
    def v_lshrrev_b64(self, d, sh, a): self._e("v_lshrrev_b64", (d, sh, a), rd=(a,), wr=(d,))
    def v_lshl_add_u64(self, d, a, sh, c): self._e("v_lshl_add_u64", (d, a, sh, c), rd=(a, c), wr=(d,))
    def v_lshl_add_u32(self, d, a, sh, c): self._e("v_lshl_add_u32", (d, a, sh, c), rd=(a, c), wr=(d,))          # (a << sh) + c
    def v_add3_u32(self, d, a, b, c): self._e("v_add3_u32", (d, a, b, c), rd=(a, b, c), wr=(d,))

    def v_mov_b32_dpp(self, d, a, ctrl, bank_mask=0xF):
        """ctrl: 'quad_perm:[0,0,2,2]' | 'row_shl:1' | 'row_shr:1' (all rows, invalid source lanes read 0); bank_mask: the BANKS of every
        row of 16 lanes that are written — bit j = lanes 4 j .. 4 j + 3 of the row (NOT lane j of every quad: the first build of the wide
        cross product read it that way, the lane model agreed with it and the GPU did not) —, the others keep what they hold"""
        self._e("v_mov_b32_dpp", (d, a), ctrl + f" row_mask:0xf bank_mask:{hex(bank_mask)} bound_ctrl:1", rd=(a,) if bank_mask == 0xF else (a, d), wr=(d,), kind="dpp")

    def v_and_b32_dpp(self, d, a, b, ctrl):
        """d = DPP(a) & b — one instruction for (cross-lane move, limb mask); b is a register"""
        self._e("v_and_b32_dpp", (d, a, b), ctrl + " row_mask:0xf bank_mask:0xf bound_ctrl:1", rd=(a, b), wr=(d,), kind="dpp")

    def v_cndmask_b32(self, d, a, b):
        """d = vcc ? b : a"""
        self._e("v_cndmask_b32_e32", (d, a, b, VCC), rd=(a, b), wr=(d,))

    def s_lshl_b64(self, d, a, sh): self._e("s_lshl_b64", (d, a, sh), rd=(a,), wr=(d,), kind="salu")

    # LDS
    def ds_read(self, bits, d, addr, offset=0):
        assert d[2] == bits // 32 and 0 <= offset < 65536 and offset % (bits // 8 if bits < 128 else 16) == 0
        self._e(f"ds_read_b{bits}", (d, addr), f"offset:{offset}" if offset else "", rd=(addr,), wr=(d,), kind="ds_read")

    def ds_write(self, bits, addr, data, offset=0):
        assert data[2] == bits // 32 and 0 <= offset < 65536
        self._e(f"ds_write_b{bits}", (addr, data), f"offset:{offset}" if offset else "", rd=(addr, data), kind="ds_write")

    def ds_add_u32(self, addr, data, offset=0):
        self._e("ds_add_u32", (addr, data), f"offset:{offset}" if offset else "", rd=(addr, data), kind="ds_write")

    # global memory
    def global_load(self, dwords, d, addr, offset=0):
        assert d[2] == dwords and addr[2] == 2 and -4096 <= offset < 4096
        name = {1: "global_load_dword", 2: "global_load_dwordx2", 4: "global_load_dwordx4"}[dwords]
        self._e(name, (d, addr, OFF), f"offset:{offset}" if offset else "", rd=(addr,), wr=(d,), kind="vm_read")

    def global_store(self, dwords, addr, data, offset=0):
        assert data[2] == dwords and addr[2] == 2 and -4096 <= offset < 4096
        name = {1: "global_store_dword", 2: "global_store_dwordx2", 4: "global_store_dwordx4"}[dwords]
        self._e(name, (addr, data, OFF), f"offset:{offset}" if offset else "", rd=(addr, data), kind="vm_write")

    # SALU / control
    def s_mov_b32(self, d, a): self._e("s_mov_b32", (d, a), rd=(a,), wr=(d,), kind="salu")
    def s_mov_b64(self, d, a): self._e("s_mov_b64", (d, a), rd=(a,), wr=(d,), kind="salu")
    def s_add_u32(self, d, a, b): self._e("s_add_u32", (d, a, b), rd=(a, b), wr=(d,), kind="salu")
    def s_sub_u32(self, d, a, b): self._e("s_sub_u32", (d, a, b), rd=(a, b), wr=(d,), kind="salu")
    def s_and_b32(self, d, a, b): self._e("s_and_b32", (d, a, b), rd=(a, b), wr=(d,), kind="salu")
    def s_cmp_lg_u32(self, a, b): self._e("s_cmp_lg_u32", (a, b), rd=(a, b), kind="salu")
    def s_cmp_eq_u32(self, a, b): self._e("s_cmp_eq_u32", (a, b), rd=(a, b), kind="salu")
    def s_bitcmp1_b32(self, a, b): self._e("s_bitcmp1_b32", (a, b), rd=(a, b), kind="salu")
    def s_cbranch_scc1(self, l): self._e("s_cbranch_scc1", (l,), kind="branch")
    def s_cbranch_scc0(self, l): self._e("s_cbranch_scc0", (l,), kind="branch")
    def s_branch(self, l): self._e("s_branch", (l,), kind="branch")
    def s_call_b64(self, d, l): self._e("s_call_b64", (d, l), wr=(d,), kind="branch")
    def s_setpc_b64(self, a): self._e("s_setpc_b64", (a,), rd=(a,), kind="branch")
    def s_and_saveexec_b64(self, d, a): self._e("s_and_saveexec_b64", (d, a), rd=(a,), wr=(d,), kind="salu")
    def s_nop(self, n): self._e("s_nop", (n,), kind="nop")

    def s_waitcnt(self, vm=None, lgkm=None):
        parts = []
        if vm is not None:
            parts.append(f"vmcnt({vm})")
        if lgkm is not None:
            parts.append(f"lgkmcnt({lgkm})")
        self._e("s_waitcnt " + " ".join(parts), (), kind="wait", comment=(vm, lgkm))

    # ---------------------------------------------------------------- text
    def lines(self):
        out = []
        for i in self.ins:
            if i.kind == "label":
                out.append(i.op)
            elif i.kind == "comment":
                out.append("  " + i.op)
            else:
                out.append("  " + i.text())
        return out

    def count(self, pred=None):
        return sum(1 for i in self.ins if i.kind not in ("label", "comment") and (pred is None or pred(i)))

    def bytes_estimate(self):
        """code size: VOP3 / DPP / literal forms and memory instructions are 8 bytes, the rest 4"""
        n = 0
        for i in self.ins:
            if i.kind in ("label", "comment"):
                continue
            big = i.op in ("v_mad_u64_u32", "v_lshrrev_b64", "v_lshl_add_u64", "v_add3_u32", "v_lshl_add_u32") or i.kind in ("dpp", "ds_read", "ds_write", "vm_read", "vm_write")
            lit = any(isinstance(o, int) and not (-16 <= o <= 64) for o in i.ops)
            n += 8 if (big or lit) else 4
        return n


# ======================================================================== the wavefront model
class Wave:
    """one wavefront: 256 VGPRs x 64 lanes, SGPRs, exec, scc, LDS (words), global memory (words by byte address / 4)"""

    def __init__(self, lds_words=40960):
        self.v = [[0] * 64 for _ in range(256)]
        self.s = [0] * 102
        self.exec = M64
        self.vcc = 0
        self.scc = 0
        self.lds = [0] * lds_words
        self.mem = {}
        self.stats = {"max_mad": 0, "instructions": 0, "valu": 0, "mad": 0}

    # operand access
    def rd32(self, o, lane):
        if isinstance(o, int):
            return o & M32
        k, i, n = o
        if k == "v":
            return self.v[i][lane]
        if k == "s":
            return self.s[i]
        raise ValueError(o)

    def rd64(self, o, lane):
        if isinstance(o, int):
            return o & M64
        k, i, n = o
        assert n == 2
        if k == "v":
            return self.v[i][lane] | (self.v[i + 1][lane] << 32)
        if k == "s":
            return self.s[i] | (self.s[i + 1] << 32)
        if k == "exec":
            return self.exec
        if k == "vcc":
            return self.vcc
        raise ValueError(o)

    def wr32(self, o, lane, x):
        assert o[0] == "v"
        self.v[o[1]][lane] = x & M32

    def wr64(self, o, lane, x):
        assert o[0] == "v" and o[2] == 2
        self.v[o[1]][lane] = x & M32
        self.v[o[1] + 1][lane] = (x >> 32) & M32

    def lanes(self):
        e = self.exec
        return [l for l in range(64) if (e >> l) & 1]


def dpp_source(ctrl, lane):
    """source lane of a DPP control, or None (reads 0 under bound_ctrl)"""
    if ctrl.startswith("quad_perm:"):
        perm = [int(x) for x in ctrl[len("quad_perm:["):-1].split(",")]
        return (lane & ~3) + perm[lane & 3]
    if ctrl == "row_shl:1":
        return lane + 1 if (lane & 15) != 15 else None
    if ctrl == "row_shr:1":
        return lane - 1 if (lane & 15) != 0 else None
    raise ValueError(ctrl)


class HazardError(AssertionError):
    pass


def run(prog, wave, entry, max_steps=10_000_000, check=True):
    """execute from label `entry` until an s_setpc_b64 through s[30:31] (the return of an entry point)"""
    ins = prog.ins
    pc = prog.labels[entry]
    RET = -1
    wave.s[30], wave.s[31] = RET & M32, 0
    # bookkeeping for the checks
    last_valu_write = {}       # vgpr -> index of the dynamic instruction that wrote it
    pending = []               # outstanding memory operations in issue order: (kind, set of destination / data vgprs)
    ds_wide = []               # (data vgprs, dynamic index) of the DS writes of more than 64 bits just issued
    dyn = 0
    steps = 0
    while True:
        steps += 1
        assert steps < max_steps, "runaway program"
        i = ins[pc]
        pc += 1
        if i.kind in ("label", "comment"):
            continue
        dyn += 1
        wave.stats["instructions"] += 1
        op = i.op
        o = i.ops
        if check:
            # destinations of loads in flight must not be touched
            busy = set()
            for kind, regs in pending:
                if kind in ("ds_read", "vm_read"):
                    busy |= regs
            for r in list(i.reads) + list(i.writes):
                if r[0] == "v" and r[1] in busy:
                    raise HazardError(f"#{pc - 1} {i.text()}: v{r[1]} is the destination of a load that was not waited for")
            if i.kind == "dpp":
                for r in i.reads:
                    if r[0] == "v" and dyn - last_valu_write.get(r[1], -10) <= 2:
                        raise HazardError(f"#{pc - 1} {i.text()}: DPP read of v{r[1]} {dyn - last_valu_write[r[1]] - 1} wait states after its VALU write (needs 2)")
            if i.kind in ("valu", "dpp"):
                for r in i.writes:
                    if r[0] == "v":
                        for regs, when in ds_wide:
                            if dyn - when <= 2 and r[1] in regs:
                                raise HazardError(f"#{pc - 1} {i.text()}: overwrites v{r[1]}, data of a DS write of more than 64 bits issued {dyn - when} slots ago")
        if i.kind in ("valu", "dpp"):
            wave.stats["valu"] += 1
            for r in i.writes:
                if r[0] == "v":
                    last_valu_write[r[1]] = dyn
        # ---------------------------------------------------------------- semantics
        if op == "v_mad_u64_u32":
            wave.stats["mad"] += 1
            d, sd, a, b, c = o
            for l in wave.lanes():
                x = wave.rd32(a, l) * wave.rd32(b, l) + wave.rd64(c, l)
                if x > wave.stats["max_mad"]:
                    wave.stats["max_mad"] = x
                assert x <= M64, f"#{pc - 1} {i.text()}: 64-bit column overflow in lane {l}"
                wave.wr64(d, l, x)
        elif op == "v_and_b32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[1], l) & wave.rd32(o[2], l))
        elif op == "v_add_u32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[1], l) + wave.rd32(o[2], l))
        elif op == "v_lshl_add_u32":
            for l in wave.lanes(): wave.wr32(o[0], l, (wave.rd32(o[1], l) << (wave.rd32(o[2], l) & 31)) + wave.rd32(o[3], l))
        elif op == "v_add3_u32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[1], l) + wave.rd32(o[2], l) + wave.rd32(o[3], l))
        elif op == "v_sub_u32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[1], l) - wave.rd32(o[2], l))
        elif op == "v_mul_u32_u24_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, (wave.rd32(o[1], l) & 0xFFFFFF) * (wave.rd32(o[2], l) & 0xFFFFFF))
        elif op == "v_lshlrev_b32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[2], l) << (wave.rd32(o[1], l) & 31))
        elif op == "v_lshrrev_b32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[2], l) >> (wave.rd32(o[1], l) & 31))
        elif op == "v_mov_b32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[1], l))
        elif op == "v_mov_b64_e32":
            for l in wave.lanes(): wave.wr64(o[0], l, wave.rd64(o[1], l))
        elif op == "v_lshrrev_b64":
            for l in wave.lanes(): wave.wr64(o[0], l, wave.rd64(o[2], l) >> (wave.rd32(o[1], l) & 63))
        elif op == "v_lshl_add_u64":
            for l in wave.lanes():
                x = (wave.rd64(o[1], l) << wave.rd32(o[2], l)) + wave.rd64(o[3], l)
                assert x <= M64, f"#{pc - 1} {i.text()}: 64-bit add overflow"
                wave.wr64(o[0], l, x)
        elif op == "v_cndmask_b32_e32":
            for l in wave.lanes(): wave.wr32(o[0], l, wave.rd32(o[2], l) if (wave.vcc >> l) & 1 else wave.rd32(o[1], l))
        elif op == "s_lshl_b64":
            x = (wave.rd64(o[1], 0) << (wave.rd32(o[2], 0) & 63)) & M64
            if o[0][0] == "vcc":
                wave.vcc = x
            else:
                wave.s[o[0][1]], wave.s[o[0][1] + 1] = x & M32, x >> 32
            wave.scc = int(x != 0)
        elif op == "v_and_b32_dpp":
            ctrl = i.mods.split(" row_mask")[0]
            src = [wave.v[o[1][1]][l] for l in range(64)]
            for l in wave.lanes():
                sl = dpp_source(ctrl, l)
                x = src[sl] if sl is not None and (wave.exec >> sl) & 1 else 0
                wave.wr32(o[0], l, x & wave.rd32(o[2], l))
        elif op == "v_mov_b32_dpp":
            ctrl = i.mods.split(" row_mask")[0]
            banks = int(i.mods.split("bank_mask:")[1].split()[0], 16)
            src = [wave.v[o[1][1]][l] for l in range(64)]
            for l in wave.lanes():
                if not (banks >> ((l & 15) >> 2)) & 1:
                    continue
                sl = dpp_source(ctrl, l)
                # (a source lane that is disabled by exec reads 0 under bound_ctrl as well)
                wave.wr32(o[0], l, src[sl] if sl is not None and (wave.exec >> sl) & 1 else 0)
        elif i.kind == "ds_read":
            n = o[0][2]
            off = int(i.mods.split(":")[1]) if i.mods else 0
            for l in wave.lanes():
                a = wave.rd32(o[1], l) + off
                assert a % (4 * min(n, 4)) == 0 or n == 4 and a % 16 == 0, f"#{pc - 1} {i.text()}: LDS address {a} misaligned"
                for j in range(n):
                    wave.v[o[0][1] + j][l] = wave.lds[a // 4 + j]
            pending.append(("ds_read", {o[0][1] + j for j in range(n)}))
        elif op.startswith("ds_write_b"):
            n = o[1][2]
            off = int(i.mods.split(":")[1]) if i.mods else 0
            for l in wave.lanes():
                a = wave.rd32(o[0], l) + off
                assert a % (4 * min(n, 4)) == 0, f"#{pc - 1} {i.text()}: LDS address {a} misaligned"
                for j in range(n):
                    wave.lds[a // 4 + j] = wave.v[o[1][1] + j][l]
            pending.append(("ds_write", set()))
            if n > 2:
                ds_wide = [(r_, w_) for (r_, w_) in ds_wide if dyn - w_ <= 2] + [({o[1][1] + j for j in range(n)}, dyn)]
        elif op == "ds_add_u32":
            off = int(i.mods.split(":")[1]) if i.mods else 0
            for l in wave.lanes():
                a = wave.rd32(o[0], l) + off
                wave.lds[a // 4] = (wave.lds[a // 4] + wave.rd32(o[1], l)) & M32
            pending.append(("ds_write", set()))
        elif i.kind == "vm_read":
            n = o[0][2]
            off = int(i.mods.split(":")[1]) if i.mods else 0
            for l in wave.lanes():
                a = wave.rd64(o[1], l) + off
                assert a % 4 == 0
                for j in range(n):
                    wave.v[o[0][1] + j][l] = wave.mem.get(a // 4 + j, 0xDEAD0000 | j)
            pending.append(("vm_read", {o[0][1] + j for j in range(n)}))
        elif i.kind == "vm_write":
            n = o[1][2]
            off = int(i.mods.split(":")[1]) if i.mods else 0
            for l in wave.lanes():
                a = wave.rd64(o[0], l) + off
                assert a % 4 == 0
                for j in range(n):
                    wave.mem[a // 4 + j] = wave.v[o[1][1] + j][l]
            pending.append(("vm_write", set()))
        elif i.kind == "wait":
            vm, lgkm = i.comment
            if lgkm is not None:
                ds = [p for p in pending if p[0].startswith("ds_")]
                keep = ds[len(ds) - lgkm:] if lgkm else []
                pending = [p for p in pending if not p[0].startswith("ds_")] + keep
            if vm is not None:
                vms = [p for p in pending if p[0].startswith("vm_")]
                keep = vms[len(vms) - vm:] if vm else []
                pending = [p for p in pending if not p[0].startswith("vm_")] + keep
        elif op == "s_mov_b32":
            if o[0][0] == "vcc_lo":
                wave.vcc = (wave.vcc & ~M32) | wave.rd32(o[1], 0)
            elif o[0][0] == "vcc_hi":
                wave.vcc = (wave.vcc & M32) | (wave.rd32(o[1], 0) << 32)
            else:
                wave.s[o[0][1]] = wave.rd32(o[1], 0)
        elif op == "s_mov_b64":
            x = wave.rd64(o[1], 0)
            if o[0][0] == "exec":
                wave.exec = x
            elif o[0][0] == "vcc":
                wave.vcc = x
            else:
                wave.s[o[0][1]], wave.s[o[0][1] + 1] = x & M32, x >> 32
        elif op == "s_add_u32":
            x = wave.rd32(o[1], 0) + wave.rd32(o[2], 0)
            wave.s[o[0][1]], wave.scc = x & M32, x >> 32
        elif op == "s_sub_u32":
            x = wave.rd32(o[1], 0) - wave.rd32(o[2], 0)
            wave.s[o[0][1]], wave.scc = x & M32, int(x < 0)
        elif op == "s_and_b32":
            x = wave.rd32(o[0 + 1], 0) & wave.rd32(o[2], 0)
            wave.s[o[0][1]], wave.scc = x, int(x != 0)
        elif op == "s_cmp_lg_u32":
            wave.scc = int(wave.rd32(o[0], 0) != wave.rd32(o[1], 0))
        elif op == "s_cmp_eq_u32":
            wave.scc = int(wave.rd32(o[0], 0) == wave.rd32(o[1], 0))
        elif op == "s_bitcmp1_b32":
            wave.scc = (wave.rd32(o[0], 0) >> (wave.rd32(o[1], 0) & 31)) & 1
        elif op == "s_and_saveexec_b64":
            old = wave.exec
            wave.s[o[0][1]], wave.s[o[0][1] + 1] = old & M32, old >> 32
            wave.exec = old & wave.rd64(o[1], 0)
            wave.scc = int(wave.exec != 0)
        elif op == "s_cbranch_scc1":
            if wave.scc: pc = prog.labels[o[0]]
        elif op == "s_cbranch_scc0":
            if not wave.scc: pc = prog.labels[o[0]]
        elif op == "s_branch":
            pc = prog.labels[o[0]]
        elif op == "s_call_b64":
            wave.s[o[0][1]], wave.s[o[0][1] + 1] = pc & M32, 0x7A000000      # (a tagged instruction index stands for the address)
            pc = prog.labels[o[1]]
        elif op == "s_setpc_b64":
            lo, hi = wave.s[o[0][1]], wave.s[o[0][1] + 1]
            if lo == (RET & M32) and hi == 0:
                if check:
                    left = [p for p in pending if p[0] in ("ds_read", "vm_read")]
                    assert not left, "entry point returns with loads in flight"
                return
            assert hi == 0x7A000000, "s_setpc_b64 through a register pair that holds no return address"
            pc = lo
        elif op == "s_nop":
            dyn += o[0]            # s_nop n: n + 1 wait states
        else:
            raise NotImplementedError(op)
