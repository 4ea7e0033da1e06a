#!/usr/bin/env python3
"""rows_ir.py — the DP row bodies of the gfx950 kernels as a tiny instruction-level IR.

One description of each algorithm's row update serves two consumers:

  * emit_asm():  gfx950 assembly text for the inline-asm row loops (gen_rows_asm.py), and
  * simulate():  a numpy interpreter of the very same instruction list, so the instruction
                 stream that ships can be checked bit-for-bit against the oracle on a CPU
                 (tests/test_rows_ir.py) before it ever runs on a GPU.

BGSA's reference emits its kernels from a generator as well (generator/source/src/main/java/org/
sduhpcl/bgsa/generator/{MyersGenerator,BitPAlGenerator}.java); this is the gfx950 counterpart.

Instruction selection follows scripts/ubench/valu_rate.hip, measured on MI355X:
  fast class (~2.2 cycles per wave64 instruction): v_and/v_or/v_xor/v_not/v_mov/v_add_u32,
      v_bitop3_b32 (any 3-input boolean), and the 4-byte VOP2 forms of v_add_co/v_addc_co that
      carry through VCC;
  slow class (~4.2 cycles, and they drag neighbouring fast instructions down with them):
      v_lshlrev/v_lshrrev, v_alignbit, v_lshl_or, v_and_or, v_or3, v_bfi, v_add3, v_bcnt, v_cmp and
      every VOP3 form that reads or writes an SGPR-pair carry.
Operand count also has a price (scripts/ubench/bank_conflict.hip): a stream of nothing but three-VGPR-
source v_bitop3 issues at ~2.95 cycles against ~2.27 for two-source VOP2 or for the two alternating;
the VGPR bank (index mod 4) of the sources makes no difference, a run of 8-byte VOP3 encodings with two
sources sits in between (~2.6).  In the 75-instruction BitPAl body of the time (49 of them three-source) turning
every v_bitop3 into a two-source instruction would buy 6.5 %, in the Myers planes body 3 % — the
instruction count is the lever, so two-source forms are used only where they cost no extra instruction.
So every 1-bit shift across words is an add-with-carry chain (x + x + carry), every chain goes
through VCC, and chains are serialised into phases.  gfx950 hazard "VALU writes VCC -> VALU reads
VCC as carry-in: 2 wait states" is enforced by the emitter (it counts the instructions between the
links of a chain and pads with s_nop if an ordering ever leaves fewer than two).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def tt(fn) -> int:
    """Truth table immediate of v_bitop3_b32 for a boolean function of (a, b, c)."""
    return fn(0xF0, 0xCC, 0xAA) & 0xFF


TT_XOR3 = tt(lambda a, b, c: a ^ b ^ c)
TT_MAJ = tt(lambda a, b, c: (a & b) | (a & c) | (b & c))
TT_OR3 = tt(lambda a, b, c: a | b | c)
TT_NOR3 = tt(lambda a, b, c: ~(a | b | c))


@dataclass
class Op:
    kind: str                 # and or xor not mov bitop3 add_co addc setc1
    dst: str = ""
    srcs: tuple = ()
    imm: int = 0
    dst2: str = ""            # second destination (shr64: the low word of the pair)


@dataclass
class Body:
    """Straight-line row body.  Register names: 'S<i>' persistent state (updated in place),
    'E<j>' the match masks of the current query character, 'P<n>' a FIXED register of the emitting loop (P<2i> / P<2i+1> = the
    low / high word of aligned pair i, for the 64-bit instructions; they keep their value from row to row), anything else is a
    temporary."""
    ops: list = field(default_factory=list)

    def _emit(self, kind, dst, *srcs, imm=0):
        self.ops.append(Op(kind, dst, tuple(srcs), imm))
        return dst

    def AND(self, d, a, b): return self._emit("and", d, a, b)
    def OR(self, d, a, b): return self._emit("or", d, a, b)
    def XOR(self, d, a, b): return self._emit("xor", d, a, b)
    def NOT(self, d, a): return self._emit("not", d, a)
    def MOV(self, d, a): return self._emit("mov", d, a)
    def BITOP3(self, d, a, b, c, fn): return self._emit("bitop3", d, a, b, c, imm=tt(fn))
    def ADD_CO(self, d, a, b): return self._emit("add_co", d, a, b)   # d = a + b, VCC = carry out
    def ADDC(self, d, a, b): return self._emit("addc", d, a, b)       # d = a + b + VCC, VCC = carry out
    def SETC1(self): return self._emit("setc1", "")                    # VCC = all ones (carry-in 1)
    # A wave has ONE fast carry register (VCC; the VOP3 forms with an SGPR-pair carry are slow class).  A chain is parked in a
    # scalar pair '$c<n>' and taken up again later (s_mov_b64: scalar pipe, no VALU issue slot) — what lets two chains run
    # over the words in turns instead of one after the other (myers_body(split=K)).
    def SAVECC(self, slot): return self._emit("savecc", f"$c{slot}")    # $c<slot> = VCC
    def LOADCC(self, slot): return self._emit("loadcc", "", f"$c{slot}")  # VCC = $c<slot>
    # The same through a VECTOR register, two fast-class VALU instructions instead of two scalar moves that wait for the vector
    # pipe: d = 0 - VCC (v_subb_co_u32 d, vcc, d, d, vcc: all ones where the carry is set; d's old value cancels), and
    # VCC = carry of d + d (v_add_co_u32 d, vcc, d, d; d is dead afterwards).
    def SAVEV(self, d): return self._emit("savev", d)                    # d = VCC ? ~0 : 0   (VCC left undefined)
    def LOADV(self, d): return self._emit("loadv", d, d)                 # VCC = d's top bit  (d clobbered)

    def ADDCZ(self, d): return self._emit("addcz", d, d)              # d = d + VCC  (v_addc_co_u32 d, vcc, 0, d, vcc)
    def SUBBZ(self, d): return self._emit("subbz", d, d)              # d = d - VCC  (v_subbrev_co_u32 d, vcc, 0, d, vcc)
    def MINU(self, d, a, b): return self._emit("minu", d, a, b)       # d = min(a, b), unsigned

    def ADD(self, d, a, b): return self._emit("add", d, a, b)          # d = a + b, no carry
    def LSHR1(self, d, a): return self._emit("lshr1", d, a)            # d = a >> 1 (slow class)
    def ALIGNBIT(self, d, hi, lo, sh): return self._emit("alignbit", d, hi, lo, sh)  # ({hi,lo} >> sh)[31:0], sh scalar
    def LSHRS(self, d, a, sh): return self._emit("lshrs", d, a, sh)   # d = a >> sh, sh scalar (VOP2 v_lshrrev_b32: fast class)
    def LSHL_IMM(self, d, a, n): return self._emit("lshl_imm", d, a, imm=n)   # d = a << n, n an immediate (slow class)

    def SHR64(self, hi, lo):
        """{hi, lo} >>= 1 as ONE v_lshrrev_b64 on an aligned register pair (P<2i+1>, P<2i>): hi's bit 0 arrives in lo's bit 31."""
        self.ops.append(Op("shr64", hi, (hi, lo), 0, lo))
        return hi

    def MATCH3(self, d, b0, b1, b2, wild=False):
        """d = columns whose 3-bit character code (b2 b1 b0) equals the row's class (0..4 = A C G T
        N): a v_bitop3 whose truth table depends on which of the five body copies it sits in.
        wild: code 7, which no character has, matches every class (the semi-global kernels' unused low
        columns) — still one instruction, only the truth tables differ."""
        tts = tuple(tt(lambda x0, x1, x2, c=c: ((x0 if c & 1 else ~x0) & (x1 if c & 2 else ~x1) & (x2 if c & 4 else ~x2))
                       | ((x0 & x1 & x2) if wild else 0))
                    for c in range(5))
        return self._emit("match3", d, b0, b1, b2, imm=tts)

    # ---- analysis ---------------------------------------------------------------------------
    def temps(self) -> list[str]:
        seen = []
        for op in self.ops:
            for r in (op.dst, op.dst2) + op.srcs:
                if r and not r.startswith(("S", "E", "B", "$", "P")) and r not in seen:
                    seen.append(r)
        return seen

    def allocate_temps(self) -> tuple[dict, int]:
        """Linear-scan reuse of temporaries: name -> slot, and the number of slots needed."""
        last_use = {}
        for i, op in enumerate(self.ops):
            for r in (op.dst,) + op.srcs:
                if r and not r.startswith(("S", "E", "B", "$", "P")):
                    last_use[r] = i
        slot_of, free, n_slots = {}, [], 0
        for i, op in enumerate(self.ops):
            # sources die at their last use, before the destination of this op is placed only if
            # the hardware reads operands before writing (it does: dst may alias a dying source)
            dying = sorted({r for r in op.srcs if r in slot_of and last_use[r] == i and r != op.dst})
            for r in dying:
                free.append(slot_of[r])
            d = op.dst
            if d and not d.startswith(("S", "E", "B", "$", "P")) and d not in slot_of:
                if free:
                    slot_of[d] = free.pop()
                else:
                    slot_of[d] = n_slots
                    n_slots += 1
            if d and d in slot_of and last_use.get(d, -1) == i:  # written, never read
                free.append(slot_of[d])
        return slot_of, n_slots

    def valu_count(self) -> int:
        return sum(op.kind not in ("setc1", "savecc", "loadcc") for op in self.ops)

    def salu_count(self) -> int:
        return sum(op.kind in ("setc1", "savecc", "loadcc") for op in self.ops)

    # ---- numpy interpreter --------------------------------------------------------------------
    def simulate(self, state: list, eq: list, cls: int = 0, planes: list | None = None,
                 scalars: dict | None = None, fixed: dict | None = None) -> None:
        """state: list of uint32 arrays (updated in place); eq: match masks of the row's class
        ('E' registers); planes: the class-independent 'B' registers; cls: the row's class; fixed: the 'P' registers, which keep
        their values between rows (the caller's dict, updated in place)."""
        regs: dict[str, np.ndarray] = {}
        fixed = {} if fixed is None else fixed
        vcc = np.zeros_like(state[0], dtype=bool)
        parked: dict[str, np.ndarray] = {}     # carry chains parked in scalar pairs (savecc / loadcc)

        def rd(name):
            if name.startswith("S"):
                return state[int(name[1:])]
            if name.startswith("E"):
                return eq[int(name[1:])]
            if name.startswith("B"):
                return planes[int(name[1:])]
            if name.startswith("$"):
                return np.full_like(state[0], scalars[name])
            if name.startswith("P"):
                return fixed[name]
            return regs[name]

        def wr(name, val):
            val = val.astype(np.uint32)
            if name.startswith("S"):
                state[int(name[1:])] = val
            elif name.startswith("P"):
                fixed[name] = val
            else:
                regs[name] = val

        FULL = np.uint32(0xFFFFFFFF)
        for op in self.ops:
            k = op.kind
            if k == "setc1":
                vcc = np.ones_like(vcc)
                continue
            if k == "savecc":
                parked[op.dst] = vcc.copy()
                continue
            if k == "loadcc":
                vcc = parked[op.srcs[0]].copy()
                continue
            if k == "savev":
                wr(op.dst, np.where(vcc, FULL, np.uint32(0)))
                vcc = vcc.copy()      # the borrow out of d - d - VCC equals VCC: the chain's carry survives, nothing relies on it
                continue
            if k == "loadv":
                d = rd(op.srcs[0])
                vcc = (d >> np.uint32(31)) != 0
                wr(op.dst, (d.astype(np.uint64) * np.uint64(2)) & np.uint64(0xFFFFFFFF))
                continue
            s = [rd(x) for x in op.srcs]
            if k == "and": wr(op.dst, s[0] & s[1])
            elif k == "or": wr(op.dst, s[0] | s[1])
            elif k == "xor": wr(op.dst, s[0] ^ s[1])
            elif k == "not": wr(op.dst, s[0] ^ FULL)
            elif k == "mov": wr(op.dst, s[0])
            elif k == "add": wr(op.dst, (s[0].astype(np.uint64) + s[1].astype(np.uint64)) & np.uint64(0xFFFFFFFF))
            elif k == "lshr1": wr(op.dst, s[0] >> np.uint32(1))
            elif k == "lshrs": wr(op.dst, s[0] >> s[1].astype(np.uint32))
            elif k == "lshl_imm": wr(op.dst, (s[0].astype(np.uint64) << np.uint64(op.imm)) & np.uint64(0xFFFFFFFF))
            elif k == "alignbit":
                pair = (s[0].astype(np.uint64) << np.uint64(32)) | s[1].astype(np.uint64)
                wr(op.dst, (pair >> s[2].astype(np.uint64)) & np.uint64(0xFFFFFFFF))
            elif k in ("bitop3", "match3"):
                a, b, c = s
                imm = op.imm if k == "bitop3" else op.imm[cls]
                out = np.zeros_like(a)
                for idx in range(8):
                    if (imm >> idx) & 1:
                        ta = a if idx & 4 else a ^ FULL
                        tb = b if idx & 2 else b ^ FULL
                        tc = c if idx & 1 else c ^ FULL
                        out |= ta & tb & tc
                wr(op.dst, out)
            elif k in ("add_co", "addc"):
                wide = s[0].astype(np.uint64) + s[1].astype(np.uint64)
                if k == "addc":
                    wide = wide + vcc.astype(np.uint64)
                vcc = (wide >> np.uint64(32)) != 0
                wr(op.dst, wide & np.uint64(0xFFFFFFFF))
            elif k == "addcz":
                wide = s[0].astype(np.uint64) + vcc.astype(np.uint64)
                vcc = (wide >> np.uint64(32)) != 0
                wr(op.dst, wide & np.uint64(0xFFFFFFFF))
            elif k == "subbz":
                borrow = vcc & (s[0] == 0)
                wr(op.dst, (s[0].astype(np.int64) - vcc.astype(np.int64)) & np.int64(0xFFFFFFFF))
                vcc = borrow
            elif k == "minu": wr(op.dst, np.minimum(s[0], s[1]))
            elif k == "shr64":
                pair = ((s[0].astype(np.uint64) << np.uint64(32)) | s[1].astype(np.uint64)) >> np.uint64(1)
                wr(op.dst, pair >> np.uint64(32))
                wr(op.dst2, pair & np.uint64(0xFFFFFFFF))
            else:
                raise ValueError(k)

    # ---- gfx950 assembly ------------------------------------------------------------------------
    def emit_asm(self, reg_name, cls: int = 0) -> list[str]:
        """reg_name(name) -> asm operand text; cls = character class of this body copy.
        Returns instruction lines with hazard padding."""
        lines: list[str] = []
        since_vcc_write = 99  # instructions issued since the last write of VCC
        for op in self.ops:
            k = op.kind
            if k == "setc1":
                lines.append("s_mov_b64 vcc, -1")
                since_vcc_write = 0
                continue
            if k == "savecc":      # a scalar read of VCC behind the VALU write: interlocked by the hardware
                lines.append(f"s_mov_b64 {reg_name(op.dst)}, vcc")
                since_vcc_write += 1
                continue
            if k == "loadcc":      # treated like setc1: its reader keeps two instructions' distance
                lines.append(f"s_mov_b64 vcc, {reg_name(op.srcs[0])}")
                since_vcc_write = 0
                continue
            if k == "savev":       # a VALU read of VCC: the chain link's two wait states
                if since_vcc_write < 2:
                    lines.append(f"s_nop {1 - since_vcc_write}")
                d = reg_name(op.dst)
                lines.append(f"v_subb_co_u32 {d}, vcc, {d}, {d}, vcc")
                since_vcc_write = 0
                continue
            if k == "loadv":
                d = reg_name(op.dst)
                lines.append(f"v_add_co_u32 {d}, vcc, {d}, {d}")
                since_vcc_write = 0
                continue
            r = [reg_name(x) for x in op.srcs]
            d = reg_name(op.dst)
            if k in ("addc", "addcz", "subbz") and since_vcc_write < 2:
                lines.append(f"s_nop {1 - since_vcc_write}")
                since_vcc_write = 2
            if k == "and": lines.append(f"v_and_b32 {d}, {r[0]}, {r[1]}")
            elif k == "or": lines.append(f"v_or_b32 {d}, {r[0]}, {r[1]}")
            elif k == "xor": lines.append(f"v_xor_b32 {d}, {r[0]}, {r[1]}")
            elif k == "not": lines.append(f"v_not_b32 {d}, {r[0]}")
            elif k == "mov": lines.append(f"v_mov_b32 {d}, {r[0]}")
            elif k == "add": lines.append(f"v_add_u32 {d}, {r[0]}, {r[1]}")
            elif k == "lshr1": lines.append(f"v_lshrrev_b32 {d}, 1, {r[0]}")
            elif k == "alignbit": lines.append(f"v_alignbit_b32 {d}, {r[0]}, {r[1]}, {r[2]}")
            elif k == "lshrs": lines.append(f"v_lshrrev_b32 {d}, {r[1]}, {r[0]}")
            elif k == "lshl_imm": lines.append(f"v_lshlrev_b32 {d}, {op.imm}, {r[0]}")
            elif k == "bitop3": lines.append(f"v_bitop3_b32 {d}, {r[0]}, {r[1]}, {r[2]} bitop3:0x{op.imm:02x}")
            elif k == "match3": lines.append(f"v_bitop3_b32 {d}, {r[0]}, {r[1]}, {r[2]} bitop3:0x{op.imm[cls]:02x}")
            elif k == "add_co": lines.append(f"v_add_co_u32 {d}, vcc, {r[0]}, {r[1]}")
            elif k == "addc": lines.append(f"v_addc_co_u32 {d}, vcc, {r[0]}, {r[1]}, vcc")
            elif k == "addcz": lines.append(f"v_addc_co_u32 {d}, vcc, 0, {r[0]}, vcc")
            elif k == "subbz": lines.append(f"v_subbrev_co_u32 {d}, vcc, 0, {r[0]}, vcc")
            elif k == "minu": lines.append(f"v_min_u32 {d}, {r[0]}, {r[1]}")
            elif k == "shr64":
                h, l = reg_name(op.srcs[0]), reg_name(op.srcs[1])
                assert h[0] == "v" and l[0] == "v" and int(h[1:]) == int(l[1:]) + 1 and int(l[1:]) % 2 == 0, (h, l)
                lines.append(f"v_lshrrev_b64 v[{l[1:]}:{h[1:]}], 1, v[{l[1:]}:{h[1:]}]")
            else:
                raise ValueError(k)
            since_vcc_write = 0 if k in ("add_co", "addc", "addcz", "subbz") else since_vcc_write + 1
        return lines


def count_hazard_nops(body: "Body") -> int:
    """s_nop padding emit_asm() would insert: addc issued fewer than 2 instructions after a VCC write."""
    nops, since = 0, 99
    for op in body.ops:
        if op.kind in ("setc1", "loadcc"):
            since = 0
            continue
        if op.kind == "savecc":
            since += 1
            continue
        if op.kind == "loadv":
            since = 0
            continue
        if op.kind in ("addc", "addcz", "subbz", "savev") and since < 2:
            nops += 1
            since = 2
        since = 0 if op.kind in ("add_co", "addc", "addcz", "subbz", "savev") else since + 1
    return nops


def schedule(body: "Body", window: int = 48) -> "Body":
    """List-schedule a straight-line body so that carry chains keep moving and the two wait states
    between the links of a chain are filled with independent work instead of s_nop.

    Dependencies: read-after-write, write-after-read and write-after-write on every register name
    and on VCC (add_co / setc1 write it, addc reads and writes it), so chains stay intact and never
    interleave.  Priority: the next chain link as soon as it is hazard-free; otherwise the earliest
    instruction in program order that does not read VCC.  `window` bounds how far ahead of the oldest
    unscheduled instruction a filler may be taken from — it trades nops against live registers."""
    ops = body.ops
    n = len(ops)
    last_write: dict = {}
    readers: dict = {}
    preds = [set() for _ in range(n)]
    for i, op in enumerate(ops):
        reads = [r for r in op.srcs if r]
        writes = [w for w in (op.dst, op.dst2) if w]
        if op.kind in ("addc", "addcz", "subbz", "savecc", "savev"):
            reads.append("VCC")
        if op.kind in ("add_co", "addc", "setc1", "addcz", "subbz", "loadcc", "savev", "loadv"):
            writes.append("VCC")
        for r in reads:
            if r in last_write:
                preds[i].add(last_write[r])
        for w in writes:
            if w in last_write:
                preds[i].add(last_write[w])
            preds[i].update(readers.get(w, ()))
        for r in reads:
            readers.setdefault(r, []).append(i)
        for w in writes:
            last_write[w] = i
            readers[w] = []
        preds[i].discard(i)
    succs = [[] for _ in range(n)]
    for i in range(n):
        for p in preds[i]:
            succs[p].append(i)
    remaining = [len(preds[i]) for i in range(n)]
    done = [False] * n
    ready = sorted(i for i in range(n) if remaining[i] == 0)
    order = []
    since = 99
    oldest = 0
    while len(order) < n:
        while oldest < n and done[oldest]:
            oldest += 1
        cands = [i for i in ready if i < oldest + window] or ready[:1]
        chain = [i for i in cands if ops[i].kind in ("addc", "add_co", "setc1", "addcz", "subbz")]
        plain = [i for i in cands if ops[i].kind not in ("addc", "addcz", "subbz")]
        if since >= 2 and chain:
            pick = chain[0]                      # the chain is the critical path: keep it moving
        elif since < 2 and plain:
            nonchain = [i for i in plain if ops[i].kind not in ("add_co", "setc1")]
            pick = (nonchain or plain)[0]        # a wait state to fill
        else:
            pick = cands[0]                      # nothing to fill it with: emit_asm() pads with s_nop
        ready.remove(pick)
        done[pick] = True
        order.append(pick)
        since = 0 if ops[pick].kind in ("add_co", "addc", "setc1", "addcz", "subbz") else since + 1
        for s in succs[pick]:
            remaining[s] -= 1
            if remaining[s] == 0:
                ready.append(s)
        ready.sort()
    out = Body()
    out.ops = [ops[i] for i in order]
    return out


def _dependencies(ops):
    """Per op the (predecessor, kind) pairs of a straight-line op list: 'raw' = reads what the predecessor wrote (a true
    dependency: the value has to exist), 'vcc' = the same through the carry register, 'order' = write-after-read /
    write-after-write (only the order matters)."""
    last_write: dict = {}
    readers: dict = {}
    deps = [[] for _ in ops]
    for i, op in enumerate(ops):
        reads = [(r, "raw") for r in op.srcs if r]
        writes = [w for w in (op.dst, op.dst2) if w]
        if op.kind in ("addc", "addcz", "subbz", "savecc", "savev"):
            reads.append(("VCC", "vcc"))
        if op.kind in ("add_co", "addc", "setc1", "addcz", "subbz", "loadcc", "savev", "loadv"):
            writes.append("VCC")
        if op.kind == "savev":
            reads = [x for x in reads if x[1] == "vcc"]      # d's old value cancels: not a dependency
        for r, kind in reads:
            if r in last_write:
                deps[i].append((last_write[r], kind))
        for w in writes:
            if w in last_write:
                deps[i].append((last_write[w], "order"))
            deps[i] += [(j, "order") for j in readers.get(w, ()) if j != i]
        for r, _ in reads:
            readers.setdefault(r, []).append(i)
        for w in writes:
            last_write[w] = i
            readers[w] = []
    return deps


def schedule_ilp(body: "Body", gap: int = 1, window: int = 24, vcc_dist: int = 3) -> "Body":
    """List-schedule a straight-line body so that no instruction issues right behind the one whose result it reads.

    Why (round 5): the Myers rows are, as written, ONE dependent chain — a = VP & E feeds the add, the add feeds HN, HN feeds M2,
    M2 feeds HP — and a gfx950 SIMD issues a wave's dependent instruction later than an independent one.  With eight waves per
    SIMD the other waves fill the slots; at two waves (the 1000 bp kernels: 228-253 VGPRs) they cannot, and every scalar
    instruction or wait in the loop shows in the time (0.877 of the issue peak against 0.93-0.98 at 150 bp).  Here every true
    dependency keeps `gap` other instructions between producer and consumer where the body has independent work to put there;
    carry links keep the two the hardware demands (VALU writes VCC -> VALU reads VCC: 2 wait states); a chain parked by SAVECC
    (a scalar read of VCC) waits `gap` + 1.  Among the instructions that may issue, the one with the longest dependent path
    behind it goes first (the carry chains are the critical path); `window` bounds how far ahead of the oldest unscheduled
    instruction one may be taken from — it trades stalls against live temporaries."""
    ops = body.ops
    n = len(ops)
    deps = _dependencies(ops)
    need = {"raw": gap + 1, "vcc": max(3, vcc_dist), "order": 1}   # vcc_dist: instructions from one carry link to the next (>= 3: two wait states)
    succs = [[] for _ in range(n)]
    for i in range(n):
        for p_, _ in deps[i]:
            succs[p_].append(i)
    height = [1] * n
    for i in range(n - 1, -1, -1):
        for s_ in succs[i]:
            height[i] = max(height[i], 1 + height[s_])
    remaining = [len({p_ for p_, _ in deps[i]}) for i in range(n)]
    pos = [None] * n
    ready = [i for i in range(n) if remaining[i] == 0]
    order = []
    oldest = 0
    done = [False] * n
    while len(order) < n:
        while oldest < n and done[oldest]:
            oldest += 1
        here = len(order)

        def earliest(i):
            t = 0
            for p_, kind in deps[i]:
                # a scalar read of VCC (SAVECC) has no wait states to keep, but it stalls its wave until the vector pipe has
                # delivered the carry: keep gap + 1 instructions between it and the link it reads
                d = gap + 2 if (ops[i].kind == "savecc" and kind == "vcc") else need[kind]
                t = max(t, pos[p_] + d)
            return t

        cands = [i for i in ready if i < oldest + window] or [min(ready)]
        free = [i for i in cands if earliest(i) <= here]
        if free:
            pick = max(free, key=lambda i: (height[i], -i))
        else:
            pick = min(cands, key=lambda i: (earliest(i), -height[i], i))
        ready.remove(pick)
        done[pick] = True
        pos[pick] = here
        order.append(pick)
        for s_ in set(succs[pick]):
            remaining[s_] -= 1
            if remaining[s_] == 0:
                ready.append(s_)
    out = Body()
    out.ops = [ops[i] for i in order]
    return out


def dependent_pairs(body: "Body") -> int:
    """How many instructions read a register the instruction right before them wrote (VALU results only)."""
    n = 0
    for prev, op in zip(body.ops, body.ops[1:]):
        if prev.kind in ("setc1", "savecc", "loadcc") or op.kind in ("setc1", "savecc", "loadcc"):
            continue
        if prev.dst and prev.dst in op.srcs:
            n += 1
    return n


# =================================================================================================
# Myers unit-cost global (reference original/BGSA_CPU/align_core.c:65-132)
# =================================================================================================

MYERS_EIGHT = __import__('os').environ.get('BGSA_GEN_MYERS_EIGHT', '1') != '0'   # '0': the 10-instruction rows of rounds 1-3 (A/B builds)


def myers_body(nw: int, groups: int = 1, split: int = 0, park: str = "sgpr", balanced: bool = False) -> Body:
    """State layout: S[(g*nw + w)*2 + 0] = VP word w of group g, +1 = VN.  E[g*nw + w] = match mask.

    Per word EIGHT instructions and two carry chains (round 4; ten and three until then: myers_body10).  In the terms of the
    general-scores body (u = 1 - dH in {0, 1, 2}: VP = [u = 0], VN = [u = 2]; v = 1 - dV likewise) a cell is m = max(W, u, v_in),
    v = m - u, new u = m - v_in with W = 2 at a match, 1 otherwise — so m is ONE bit, M2 = [m = 2] = E | VN | [v_in = 2], and
    [v_in = 2] of column j + 1 is VP_j & M2_j = (VP_j & E_j) | (VP_j & [v_in = 2]_j): a carry with generate VP & E and propagate
    VP, i.e. the carries of the addition VP + (VP & E), and the carry INTO a bit is sum ^ VP ^ (VP & E) — [v_in = 2] comes out of
    the adder already moved up one column, across the words.  What the classical order pays for (HN and its shift) is gone:
        a = VP & E ; s = VP + a (chain, carry-in 0) ; HNs = s ^ VP ^ a ; M2 = E | VN | HNs ; HP = VN | ~(M2 | VP)
        HPs = HP << 1 (chain, carry-in 1: the row edge) ; VN' = M2 & HPs ; VP' = (M2 & HNs) | ~(M2 | HPs)
    (reference: 24 ops per 31-bit word, align_core.c:72-103).  Two temporaries per word: HP is formed in the dead VP register.

    split = K > 0 (round 5): the two chains take turns over blocks of K words instead of running over all the words one after
    the other — phase A of words 0..K-1, phase B of the same words, phase A of K..2K-1, ... — with the chain that pauses parked
    in a scalar pair (SAVECC / LOADCC, scalar pipe).  The two temporaries of a word then live for one block, not for the row:
    2K temporaries instead of 2*nw, which is what lets 30 and 32 words keep their five Peq planes resident (7*nw + 2K + the
    kernel's own registers <= 256) where they ran on the code planes, nine instructions per word, before.
    park = "vgpr": the pausing chain waits in a vector register (SAVEV / LOADV: one fast-class VALU instruction each, two more
    temporaries) instead of a scalar pair — a scalar move of VCC waits for the vector pipe to drain, and at two waves per
    SIMD nobody fills the gap.
    balanced: HP of word w + 1 is formed in phase B, between the shift of word w and its new deltas, instead of at the end of
    phase A — four instructions from carry link to carry link in BOTH phases where the order above has five and three."""
    if not MYERS_EIGHT:
        return myers_body10(nw, groups)
    b = Body()
    for g in range(groups):
        VP = lambda w: f"S{(g * nw + w) * 2}"
        VN = lambda w: f"S{(g * nw + w) * 2 + 1}"
        E = lambda w: f"E{g * nw + w}"
        A = lambda w: f"a{g}_{w}"      # VP & E, then [v_in = 2]
        M = lambda w: f"m{g}_{w}"      # the sum, then M2
        blocks = [(0, nw)] if split <= 0 else [(lo, min(nw, lo + split)) for lo in range(0, nw, split)]
        vg = park == "vgpr"
        save = (lambda slot: b.SAVEV(f"cv{g}_{slot}")) if vg else b.SAVECC
        load = (lambda slot: b.LOADV(f"cv{g}_{slot}")) if vg else b.LOADCC
        for j, (lo, hi) in enumerate(blocks):
            first, last = j == 0, j == len(blocks) - 1
            for w in range(lo, hi):  # phase A: the addition's carry chain
                b.AND(A(w), VP(w), E(w))
                (b.ADD_CO if w == 0 else b.ADDC)(M(w), VP(w), A(w))
                if w == hi - 1 and not last and not vg:
                    save(0)
                if w == hi - 1 and not vg and balanced and hi - lo > 1:
                    b.SETC1() if first else load(1)      # balanced: only two instructions of this phase are left behind it
                b.BITOP3(A(w), M(w), VP(w), A(w), lambda s_, vp, a: s_ ^ vp ^ a)
                if w == hi - 1 and not vg and not (balanced and hi - lo > 1):
                    # the other chain's carry, two instructions ahead of its reader; block 0: the row edge D[i][0] - D[i-1][0] = +1
                    # enters HP at bit 0 of word 0
                    b.SETC1() if first else load(1)
                b.BITOP3(M(w), E(w), VN(w), A(w), lambda e, vn, hn: e | vn | hn)
                if w == hi - 1 and not last and vg:
                    save(0)          # a VALU read of VCC: two instructions behind the link that wrote it
                if not balanced or w == lo:
                    b.BITOP3(VP(w), M(w), VN(w), VP(w), lambda m, vn, vp: (m & vn) | ~(m | vp))
                if w == hi - 1 and vg:
                    b.SETC1() if first else load(1)     # (as written its reader follows at once: schedule_ilp, or the emitter's s_nop, spaces them)
            for w in range(lo, hi):  # phase B: HP << 1 across words, then the new deltas
                b.ADDC(VP(w), VP(w), VP(w))
                if balanced and w + 1 < hi:
                    b.BITOP3(VP(w + 1), M(w + 1), VN(w + 1), VP(w + 1), lambda m, vn, vp: (m & vn) | ~(m | vp))
                if w == hi - 1 and not last and not vg:
                    save(1)
                b.AND(VN(w), M(w), VP(w))
                if w == hi - 1 and not last and not vg:
                    load(0)
                b.BITOP3(VP(w), M(w), A(w), VP(w), lambda m, hn, hp: (m & hn) | ~(m | hp))
                if w == hi - 1 and not last and vg:
                    save(1)
                    load(0)
    return b


def myers_body10(nw: int, groups: int = 1) -> Body:
    """State layout: S[(g*nw + w)*2 + 0] = VP word w of group g, +1 = VN.  E[g*nw + w] = match mask.

    Per word (10 instructions, full 32-bit words, hardware carries instead of the reference's
    software carry bit, align_core.c:79-83,91-96):
        D0 = (((P & E) + P) ^ P) | E | M ; HP = ~(D0 | P) | M ; HN = D0 & P
        HP, HN <<= 1 (row boundary feeds 1 into HP)  ; P' = ~(D0 | HP) | HN ; M' = D0 & HP
    """
    b = Body()
    for g in range(groups):
        P = lambda w: f"S{(g * nw + w) * 2}"
        M = lambda w: f"S{(g * nw + w) * 2 + 1}"
        E = lambda w: f"E{g * nw + w}"
        D = lambda w: f"d{g}_{w}"
        HP = lambda w: f"hp{g}_{w}"
        HN = lambda w: f"hn{g}_{w}"
        for w in range(nw):  # phase A: the addition's carry chain
            b.AND(D(w), P(w), E(w))   # (P & (E|M)) == (P & E): P & M == 0 is an invariant
            (b.ADD_CO if w == 0 else b.ADDC)(D(w), D(w), P(w))
            b.BITOP3(D(w), D(w), P(w), M(w), lambda a, p, m: (a ^ p) | m)
            b.OR(D(w), D(w), E(w))
        b.SETC1()  # D[i][0] - D[i-1][0] = +1 enters HP at bit 0 of word 0
        for w in range(nw):  # phase C: HP << 1 across words
            b.BITOP3(HP(w), D(w), P(w), M(w), lambda d, p, m: ~(d | p) | m)
            b.AND(HN(w), D(w), P(w))
            b.ADDC(HP(w), HP(w), HP(w))
        for w in range(nw):  # phase D: HN << 1 across words, then the new vertical deltas
            (b.ADD_CO if w == 0 else b.ADDC)(HN(w), HN(w), HN(w))
            b.AND(M(w), D(w), HP(w))
            b.BITOP3(P(w), D(w), HP(w), HN(w), lambda d, hp, hn: ~(d | hp) | hn)
    return b


def myers_parked_body(nw: int) -> Body:
    """myers_body for 26..28 words: HN is parked in the VP register between the phases (as myers_planes_body does), so
    a word needs 9 registers instead of 10 — 5 Peq masks, VP, VN, D0, HP — and 28 words (896 bp) still fit 256 VGPRs
    with two waves per SIMD.  Same 10 instructions per word; state and masks as myers_body (one group).
    (Round 4's eight-instruction myers_body needs nine registers per word as it is — HP lives in the dead VP register — and replaces it.)"""
    if MYERS_EIGHT:
        return myers_body(nw, 1)
    b = Body()
    P = lambda w: f"S{w * 2}"
    M = lambda w: f"S{w * 2 + 1}"
    E = lambda w: f"E{w}"
    D = lambda w: f"d{w}"
    HP = lambda w: f"hp{w}"
    for w in range(nw):  # phase A
        b.AND(D(w), P(w), E(w))
        (b.ADD_CO if w == 0 else b.ADDC)(D(w), D(w), P(w))
        b.BITOP3(D(w), D(w), P(w), M(w), lambda a, p, m: (a ^ p) | m)
        b.OR(D(w), D(w), E(w))
    b.SETC1()
    for w in range(nw):  # phase C: HP chain; HN parks in the VP register
        b.BITOP3(HP(w), D(w), P(w), M(w), lambda d, p, m: ~(d | p) | m)
        b.AND(P(w), D(w), P(w))
        b.ADDC(HP(w), HP(w), HP(w))
    for w in range(nw):  # phase D: HN chain in place, then the new VP
        (b.ADD_CO if w == 0 else b.ADDC)(P(w), P(w), P(w))
        b.AND(M(w), D(w), HP(w))
        b.BITOP3(P(w), D(w), HP(w), P(w), lambda d, hp, hn: ~(d | hp) | hn)
    return b


def myers_semi_body(nw: int, split: int = 0) -> Body:
    """Semi-global Myers (generator -m 0 -s, MyersGenerator.java:56-223): the subject end to end inside the
    query — D[i][0] = 0 for every query row i, result = min over i of D[i][n].  The body is myers_body with
      * carry-in 0 instead of 1 for the HP shift (the row edge D[i][0] - D[i-1][0] = 0), and
      * the subject RIGHT-ALIGNED in its words (the kernel shifts the masks once per task; the unused low
        columns match every character and start at VP = 0, so they stay at D = 0 and hand the row edge to
        the first real column), which puts the last subject column at bit 31 of the last word: the carries
        that leave the HP and HN shift chains ARE D[i][n] - D[i-1][n], added to / subtracted from the running
        score with one add-with-carry each, plus one v_min for the best score — 3 instructions per row on top of
        the 10 per word, nothing extracted bit by bit.
    State: myers_body's, then S[2nw] = D[i][n] (running), S[2nw+1] = its minimum so far.
    split = K > 0: the two chains in turns over blocks of K words, as myers_body(split=K) — 2K temporaries, which keeps the five Peq
    planes resident up to 32 words (round 5); the two carries that leave the chains are taken where the LAST block's phases end."""
    if MYERS_EIGHT and split > 0:
        b = Body()
        VP = lambda w: f"S{w * 2}"
        VN = lambda w: f"S{w * 2 + 1}"
        RUN, BEST = f"S{2 * nw}", f"S{2 * nw + 1}"
        blocks = [(lo, min(nw, lo + split)) for lo in range(0, nw, split)]
        for j, (lo, hi) in enumerate(blocks):
            first, last = j == 0, j == len(blocks) - 1
            for w in range(lo, hi):      # phase A: the addition's chain
                b.AND(f"a{w}", VP(w), f"E{w}")
                (b.ADD_CO if w == 0 else b.ADDC)(f"m{w}", VP(w), f"a{w}")
                if w == hi - 1 and not last:
                    b.SAVECC(0)
                b.BITOP3(f"a{w}", f"m{w}", VP(w), f"a{w}", lambda s_, vp, a_: s_ ^ vp ^ a_)
                b.BITOP3(f"m{w}", f"E{w}", VN(w), f"a{w}", lambda e, vn, hn: e | vn | hn)
                if w == hi - 1 and last:
                    b.SUBBZ(RUN)             # - the HN bit that left the last column (two instructions behind the chain's last link)
                if w == hi - 1 and not first:
                    b.LOADCC(1)              # the HP shift's carry out of the block before (block 0 starts it with carry-in 0: ADD_CO)
                b.BITOP3(VP(w), f"m{w}", VN(w), VP(w), lambda m, vn, vp: (m & vn) | ~(m | vp))
            for w in range(lo, hi):      # phase B: HP << 1 across words
                (b.ADD_CO if w == 0 else b.ADDC)(VP(w), VP(w), VP(w))
                if w == hi - 1 and not last:
                    b.SAVECC(1)
                b.AND(VN(w), f"m{w}", VP(w))
                if w == hi - 1 and last:
                    b.ADDCZ(RUN)             # + the HP bit that left the last column (before anything else writes VCC)
                if w == hi - 1 and not last:
                    b.LOADCC(0)
                b.BITOP3(VP(w), f"m{w}", f"a{w}", VP(w), lambda m, hn, hp: (m & hn) | ~(m | hp))
        b.MINU(BEST, BEST, RUN)
        return b
    if MYERS_EIGHT:
        # the eight-instruction row (myers_body): the carry that LEAVES the addition chain is [v = 2] of the last column = its HN bit,
        # the one that leaves the HP shift its HP bit — the same two score updates, 8 VALU per word + 3 per row
        b = Body()
        VP = lambda w: f"S{w * 2}"
        VN = lambda w: f"S{w * 2 + 1}"
        RUN, BEST = f"S{2 * nw}", f"S{2 * nw + 1}"
        for w in range(nw):
            b.AND(f"a{w}", VP(w), f"E{w}")
            (b.ADD_CO if w == 0 else b.ADDC)(f"m{w}", VP(w), f"a{w}")
            b.BITOP3(f"a{w}", f"m{w}", VP(w), f"a{w}", lambda s_, vp, a_: s_ ^ vp ^ a_)
            b.BITOP3(f"m{w}", f"E{w}", VN(w), f"a{w}", lambda e, vn, hn: e | vn | hn)
            if w == nw - 1:
                b.SUBBZ(RUN)                    # - the HN bit that left the last column (two instructions behind the chain's last link)
            b.BITOP3(VP(w), f"m{w}", VN(w), VP(w), lambda m, vn, vp: (m & vn) | ~(m | vp))
        for w in range(nw):  # HP << 1 across words, carry-in 0
            (b.ADD_CO if w == 0 else b.ADDC)(VP(w), VP(w), VP(w))
            b.AND(VN(w), f"m{w}", VP(w))
            if w == nw - 1:
                b.ADDCZ(RUN)                    # + the HP bit that left the last column (before anything else writes VCC)
            b.BITOP3(VP(w), f"m{w}", f"a{w}", VP(w), lambda m, hn, hp: (m & hn) | ~(m | hp))
        b.MINU(BEST, BEST, RUN)
        return b
    b = Body()
    P = lambda w: f"S{w * 2}"
    M = lambda w: f"S{w * 2 + 1}"
    E = lambda w: f"E{w}"
    D = lambda w: f"d{w}"
    HP = lambda w: f"hp{w}"
    HN = lambda w: f"hn{w}"
    RUN, BEST = f"S{2 * nw}", f"S{2 * nw + 1}"
    for w in range(nw):
        b.AND(D(w), P(w), E(w))
        (b.ADD_CO if w == 0 else b.ADDC)(D(w), D(w), P(w))
        b.BITOP3(D(w), D(w), P(w), M(w), lambda a, p, m: (a ^ p) | m)
        b.OR(D(w), D(w), E(w))
    for w in range(nw):  # HP << 1 across words, carry-in 0
        b.BITOP3(HP(w), D(w), P(w), M(w), lambda d, p, m: ~(d | p) | m)
        b.AND(HN(w), D(w), P(w))
        (b.ADD_CO if w == 0 else b.ADDC)(HP(w), HP(w), HP(w))
    b.AND(M(0), D(0), HP(0))            # two instructions behind the chain's last link
    if nw > 1:
        b.AND(M(1), D(1), HP(1))
    b.ADDCZ(RUN)                        # + the HP bit that left the last column
    for w in range(nw):  # HN << 1 across words, then the new vertical deltas
        (b.ADD_CO if w == 0 else b.ADDC)(HN(w), HN(w), HN(w))
        if w >= 2:
            b.AND(M(w), D(w), HP(w))
        b.BITOP3(P(w), D(w), HP(w), HN(w), lambda d, hp, hn: ~(d | hp) | hn)
    b.SUBBZ(RUN)                        # - the HN bit that left the last column
    b.MINU(BEST, BEST, RUN)
    return b


def semi_align(peq: np.ndarray, slen: int, nw: int):
    """What the semi-global kernels do to a subject's match masks once per task: right-align the slen columns
    in nw words and make the unused low columns match everything.  Returns (aligned peq [5][nw][n], VP init
    per word [nw] — 0 in the unused columns, 1 in the subject's)."""
    s = 32 * nw - slen
    assert s >= 0
    n = peq.shape[2]
    out = np.zeros((5, nw, n), dtype=np.uint32)
    vp = []
    for w in range(nw):
        lo_col = 32 * w - s                    # source column of bit 0 of aligned word w
        word = np.zeros((5, n), dtype=np.uint64)
        for src in (lo_col // 32, lo_col // 32 + 1):
            if 0 <= src < peq.shape[1]:
                shift = 32 * src - lo_col      # where source word `src` starts within the aligned word
                v = peq[:, src].astype(np.uint64)
                word |= (v << np.uint64(shift)) if shift >= 0 else (v >> np.uint64(-shift))
        dummy = 0xFFFFFFFF if 32 * (w + 1) <= s else ((1 << (s - 32 * w)) - 1 if 32 * w < s else 0)
        out[:, w] = ((word & np.uint64(0xFFFFFFFF)) | np.uint64(dummy)).astype(np.uint32)
        vp.append(np.uint32(~dummy & 0xFFFFFFFF))
    return out, vp


def myers_semi_simulate(subjects: np.ndarray, query: np.ndarray, nw: int, body: "Body | None" = None) -> np.ndarray:
    """One query against the subjects with myers_semi_body (or the given form of it), set up as myers_semi_asm_kernel<NW> does."""
    n, slen = subjects.shape
    peq, vp0 = semi_align(build_peq32(subjects, (slen + 31) // 32), slen, nw)
    st = []
    for w in range(nw):
        st += [np.full(n, vp0[w], np.uint32), np.zeros(n, np.uint32)]
    st += [np.full(n, slen, np.uint32), np.full(n, slen, np.uint32)]     # D[0][n] = n
    body = myers_semi_body(nw) if body is None else body
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
    for ch in query:
        c = code.get(int(ch), 0)
        body.simulate(st, [peq[c, w] for w in range(nw)])
    return (-st[2 * nw + 1].astype(np.int64)).astype(np.int16)


def myers_planes_body(nw: int, split: int = 0, balanced: bool = False) -> Body:
    """Myers for long subjects (769..1024 bp): the five Peq planes of a subject (5*nw registers)
    are replaced by its 3-bit character code planes B[w*3+i] (3*nw registers) and the match mask
    of the row's class is rebuilt per word with one v_bitop3 (MATCH3) — 11 instructions per word,
    7*nw+1 registers, which keeps 32 words (1024 bp) at two waves per SIMD.  State as myers_body.
    split = K > 0: the chains in turns over blocks of K words (see myers_body): 5*nw + 2K + 1 registers.
    """
    if MYERS_EIGHT:      # myers_body's eight instructions + the match mask: 9 per word, the same 7*nw + 1 registers
        b = Body()
        VP = lambda w: f"S{w * 2}"
        VN = lambda w: f"S{w * 2 + 1}"
        blocks = [(0, nw)] if split <= 0 else [(lo, min(nw, lo + split)) for lo in range(0, nw, split)]
        for j, (lo, hi) in enumerate(blocks):
            first, last = j == 0, j == len(blocks) - 1
            for w in range(lo, hi):
                b.MATCH3(f"e{w}", f"B{w * 3}", f"B{w * 3 + 1}", f"B{w * 3 + 2}")   # a name per word: no false dependency between the words' masks (the slots are reused)
                b.AND(f"a{w}", VP(w), f"e{w}")
                (b.ADD_CO if w == 0 else b.ADDC)(f"m{w}", VP(w), f"a{w}")
                if w == hi - 1 and not last:
                    b.SAVECC(0)
                if w == nw - 1 and not split:
                    b.SETC1()
                b.BITOP3(f"a{w}", f"m{w}", VP(w), f"a{w}", lambda s_, vp, a_: s_ ^ vp ^ a_)
                if w == hi - 1 and split:
                    b.SETC1() if first else b.LOADCC(1)
                b.BITOP3(f"m{w}", f"e{w}", VN(w), f"a{w}", lambda e, vn, hn: e | vn | hn)
                if not balanced or w == lo:
                    b.BITOP3(VP(w), f"m{w}", VN(w), VP(w), lambda m, vn, vp: (m & vn) | ~(m | vp))
            for w in range(lo, hi):
                b.ADDC(VP(w), VP(w), VP(w))
                if balanced and w + 1 < hi:   # HP of the next word between this word's shift and its new deltas (myers_body: balanced)
                    b.BITOP3(VP(w + 1), f"m{w + 1}", VN(w + 1), VP(w + 1), lambda m, vn, vp: (m & vn) | ~(m | vp))
                if w == hi - 1 and not last:
                    b.SAVECC(1)
                b.AND(VN(w), f"m{w}", VP(w))
                if w == hi - 1 and not last:
                    b.LOADCC(0)
                b.BITOP3(VP(w), f"m{w}", f"a{w}", VP(w), lambda m, hn, hp: (m & hn) | ~(m | hp))
        return b
    b = Body()
    P = lambda w: f"S{w * 2}"
    M = lambda w: f"S{w * 2 + 1}"
    D = lambda w: f"d{w}"
    HP = lambda w: f"hp{w}"
    for w in range(nw):  # phase A
        b.MATCH3("e", f"B{w * 3}", f"B{w * 3 + 1}", f"B{w * 3 + 2}")
        b.AND(D(w), P(w), "e")
        (b.ADD_CO if w == 0 else b.ADDC)(D(w), D(w), P(w))
        b.BITOP3(D(w), D(w), P(w), M(w), lambda a, p, m: (a ^ p) | m)
        b.OR(D(w), D(w), "e")
    b.SETC1()
    for w in range(nw):  # phase C: HP chain; HN parks in the VP register
        b.BITOP3(HP(w), D(w), P(w), M(w), lambda d, p, m: ~(d | p) | m)
        b.AND(P(w), D(w), P(w))            # HN
        b.ADDC(HP(w), HP(w), HP(w))
    for w in range(nw):  # phase D: HN chain in place, then the new VP; the new VN (needs only D and the
        #                  shifted HP) is computed here so that every link has two instructions behind it
        (b.ADD_CO if w == 0 else b.ADDC)(P(w), P(w), P(w))
        b.AND(M(w), D(w), HP(w))
        b.BITOP3(P(w), D(w), HP(w), P(w), lambda d, hp, hn: ~(d | hp) | hn)
    return b


def myers_semi_planes_body(nw: int) -> Body:
    """Semi-global Myers for subjects of 769..1024 bp: myers_planes_body with myers_semi_body's changes — HP shift
    with carry-in 0, the subject right-aligned, the carries leaving the HP / HN chains accumulated into the running
    last-column score — and the unused low columns coded 7 in the planes, the code that matches every class (MATCH3
    wild), so they cost no instruction.  11 VALU per word + 3 per row.
    State: S[2w] = VP, S[2w+1] = VN, S[2nw] = D[i][n], S[2nw+1] = its minimum so far."""
    if MYERS_EIGHT:      # as myers_semi_body's eight-instruction form, the match mask from the code planes: 9 per word + 3 per row
        b = Body()
        VP = lambda w: f"S{w * 2}"
        VN = lambda w: f"S{w * 2 + 1}"
        RUN, BEST = f"S{2 * nw}", f"S{2 * nw + 1}"
        for w in range(nw):
            b.MATCH3("e", f"B{w * 3}", f"B{w * 3 + 1}", f"B{w * 3 + 2}", wild=True)
            b.AND(f"a{w}", VP(w), "e")
            (b.ADD_CO if w == 0 else b.ADDC)(f"m{w}", VP(w), f"a{w}")
            b.BITOP3(f"a{w}", f"m{w}", VP(w), f"a{w}", lambda s_, vp, a_: s_ ^ vp ^ a_)
            b.BITOP3(f"m{w}", "e", VN(w), f"a{w}", lambda e, vn, hn: e | vn | hn)
            if w == nw - 1:
                b.SUBBZ(RUN)
            b.BITOP3(VP(w), f"m{w}", VN(w), VP(w), lambda m, vn, vp: (m & vn) | ~(m | vp))
        for w in range(nw):
            (b.ADD_CO if w == 0 else b.ADDC)(VP(w), VP(w), VP(w))
            b.AND(VN(w), f"m{w}", VP(w))
            if w == nw - 1:
                b.ADDCZ(RUN)
            b.BITOP3(VP(w), f"m{w}", f"a{w}", VP(w), lambda m, hn, hp: (m & hn) | ~(m | hp))
        b.MINU(BEST, BEST, RUN)
        return b
    b = Body()
    P = lambda w: f"S{w * 2}"
    M = lambda w: f"S{w * 2 + 1}"
    D = lambda w: f"d{w}"
    HP = lambda w: f"hp{w}"
    RUN, BEST = f"S{2 * nw}", f"S{2 * nw + 1}"
    for w in range(nw):
        b.MATCH3("e", f"B{w * 3}", f"B{w * 3 + 1}", f"B{w * 3 + 2}", wild=True)
        b.AND(D(w), P(w), "e")
        (b.ADD_CO if w == 0 else b.ADDC)(D(w), D(w), P(w))
        b.BITOP3(D(w), D(w), P(w), M(w), lambda a, p, m: (a ^ p) | m)
        b.OR(D(w), D(w), "e")
    for w in range(nw):  # HP chain, carry-in 0; HN parks in the VP register
        b.BITOP3(HP(w), D(w), P(w), M(w), lambda d, p, m: ~(d | p) | m)
        b.AND(P(w), D(w), P(w))
        (b.ADD_CO if w == 0 else b.ADDC)(HP(w), HP(w), HP(w))
    b.AND(M(0), D(0), HP(0))            # two instructions behind the chain's last link
    b.AND(M(1), D(1), HP(1))
    b.ADDCZ(RUN)                        # + the HP bit that left the last column
    for w in range(nw):  # HN chain in place, then the new vertical deltas
        (b.ADD_CO if w == 0 else b.ADDC)(P(w), P(w), P(w))
        if w >= 2:
            b.AND(M(w), D(w), HP(w))
        b.BITOP3(P(w), D(w), HP(w), P(w), lambda d, hp, hn: ~(d | hp) | hn)
    b.SUBBZ(RUN)                        # - the HN bit that left the last column
    b.MINU(BEST, BEST, RUN)
    return b


def myers_semi_planes_simulate(subjects: np.ndarray, query: np.ndarray, nw: int) -> np.ndarray:
    """One query against the subjects with myers_semi_planes_body, set up as myers_semi_planes_kernel<NW> does."""
    n, slen = subjects.shape
    wn = (slen + 31) // 32
    peq = build_peq32(subjects, wn)
    planes5 = [peq[1] | peq[3], peq[2] | peq[3], peq[4]]                 # [3][wn][n]
    s = 32 * nw - slen
    B, st = [], []
    for w in range(nw):
        lo_col = 32 * w - s
        dummy = 0xFFFFFFFF if 32 * (w + 1) <= s else ((1 << (s - 32 * w)) - 1 if 32 * w < s else 0)
        for i in range(3):
            word = np.zeros(n, dtype=np.uint64)
            for src in (lo_col // 32, lo_col // 32 + 1):
                if 0 <= src < wn:
                    shift = 32 * src - lo_col
                    v = planes5[i][src].astype(np.uint64)
                    word |= (v << np.uint64(shift)) if shift >= 0 else (v >> np.uint64(-shift))
            B.append(((word & np.uint64(0xFFFFFFFF)) | np.uint64(dummy)).astype(np.uint32))
        st += [np.full(n, ~dummy & 0xFFFFFFFF, np.uint32), np.zeros(n, np.uint32)]
    st += [np.full(n, slen, np.uint32), np.full(n, slen, np.uint32)]
    body = myers_semi_planes_body(nw)
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
    for ch in query:
        body.simulate(st, [], cls=code.get(int(ch), 0), planes=B)
    return (-st[2 * nw + 1].astype(np.int64)).astype(np.int16)


def myers_block_body(nw: int) -> Body:
    """myers_planes_body for one column block of a subject longer than 1024 bp.  The three carry
    chains enter and leave the block through registers that hold 32 rows' worth of carry bits,
    first row in bit 31:  x + x  shifts the next row's carry-in into VCC,  x + x + vcc  appends
    the row's carry-out — six extra fast-class instructions per row, no bit extraction.
    State: S[2w] = VP, S[2w+1] = VN, then CIN_S, CIN_P, CIN_N, COUT_S, COUT_P, COUT_N."""
    if MYERS_EIGHT:      # two chains: the addition (its carries are [v_in = 2]) and the HP shift; the third pair of carry words stays unused
        b = Body()
        VP = lambda w: f"S{w * 2}"
        VN = lambda w: f"S{w * 2 + 1}"
        CIN = [f"S{2 * nw + i}" for i in range(3)]
        COUT = [f"S{2 * nw + 3 + i}" for i in range(3)]
        b.ADD_CO(CIN[0], CIN[0], CIN[0])
        for w in range(nw):
            b.MATCH3("e", f"B{w * 3}", f"B{w * 3 + 1}", f"B{w * 3 + 2}")
            b.AND(f"a{w}", VP(w), "e")
            b.ADDC(f"m{w}", VP(w), f"a{w}")
            if w == nw - 1:
                b.ADDC(COUT[0], COUT[0], COUT[0])
                b.ADD_CO(CIN[1], CIN[1], CIN[1])
            b.BITOP3(f"a{w}", f"m{w}", VP(w), f"a{w}", lambda s_, vp, a_: s_ ^ vp ^ a_)
            b.BITOP3(f"m{w}", "e", VN(w), f"a{w}", lambda e, vn, hn: e | vn | hn)
            b.BITOP3(VP(w), f"m{w}", VN(w), VP(w), lambda m, vn, vp: (m & vn) | ~(m | vp))
        for w in range(nw):
            b.ADDC(VP(w), VP(w), VP(w))
            b.AND(VN(w), f"m{w}", VP(w))
            b.BITOP3(VP(w), f"m{w}", f"a{w}", VP(w), lambda m, hn, hp: (m & hn) | ~(m | hp))
        b.ADDC(COUT[1], COUT[1], COUT[1])
        return schedule(b, 16)
    b = Body()
    P = lambda w: f"S{w * 2}"
    M = lambda w: f"S{w * 2 + 1}"
    D = lambda w: f"d{w}"
    HP = lambda w: f"hp{w}"
    CIN = [f"S{2 * nw + i}" for i in range(3)]
    COUT = [f"S{2 * nw + 3 + i}" for i in range(3)]
    b.ADD_CO(CIN[0], CIN[0], CIN[0])
    for w in range(nw):
        b.MATCH3("e", f"B{w * 3}", f"B{w * 3 + 1}", f"B{w * 3 + 2}")
        b.AND(D(w), P(w), "e")
        b.ADDC(D(w), D(w), P(w))
        b.BITOP3(D(w), D(w), P(w), M(w), lambda a, p, m: (a ^ p) | m)
        b.OR(D(w), D(w), "e")
    b.ADDC(COUT[0], COUT[0], COUT[0])
    b.ADD_CO(CIN[1], CIN[1], CIN[1])
    for w in range(nw):
        b.BITOP3(HP(w), D(w), P(w), M(w), lambda d, p, m: ~(d | p) | m)
        b.AND(P(w), D(w), P(w))            # HN parks in the VP register
        b.ADDC(HP(w), HP(w), HP(w))
    b.ADDC(COUT[1], COUT[1], COUT[1])
    b.ADD_CO(CIN[2], CIN[2], CIN[2])
    for w in range(nw):
        b.ADDC(P(w), P(w), P(w))
        b.AND(M(w), D(w), HP(w))
        b.BITOP3(P(w), D(w), HP(w), P(w), lambda d, hp, hn: ~(d | hp) | hn)
    b.ADDC(COUT[2], COUT[2], COUT[2])
    return schedule(b, 16)   # fills the slots around the carry-word instructions


def myers_peq_block_body(nw: int) -> Body:
    """Column-block form of myers_body (Peq planes resident, 8 VALU per word + 4 for the carry words; 10 + 6 until round 4),
    derived mechanically like the BitPAl block bodies.  Chains: addition (carry-in 0), HP shift (carry-in
    1 in the first block: the row edge), HN shift (0)."""
    if MYERS_EIGHT:      # two chains in the first two of the kernel's three carry-word pairs (the second starts with the row edge, as before)
        body, init = make_blocked(myers_body(nw, 1), 2 * nw, n_slots=3)
        assert init == [0, 1]
        return schedule(body, 16)
    body, init = make_blocked(myers_body10(nw, 1), 2 * nw)
    assert init == [0, 1, 0]
    return schedule(body, 16)


def myers_blocked_simulate(subjects: np.ndarray, query: np.ndarray, nw_block: int, peq_resident: bool = False) -> np.ndarray:
    """A subject processed as column blocks of nw_block words with the block body, carries
    exchanged through per-32-row words exactly as myers_blocked_kernel does.  Returns int16.
    peq_resident: the block body built on myers_body (match masks in registers) instead of the code planes."""
    n, slen = subjects.shape
    qlen = len(query)
    nw_total = (slen + 31) // 32
    n_blocks = (nw_total + nw_block - 1) // nw_block
    peq = build_peq32(subjects, n_blocks * nw_block)      # zero masks beyond the subject
    planes_all = code_planes(peq)
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
    n_chunks = (qlen + 31) // 32
    FULL = np.uint32(0xFFFFFFFF)
    # carry words per 32-row chunk: addition carry-in 0, HP carry-in 1 (row edge), HN carry-in 0
    carry = [[np.zeros(n, np.uint32), np.full(n, FULL, np.uint32), np.zeros(n, np.uint32)] for _ in range(n_chunks)]
    body = myers_peq_block_body(nw_block) if peq_resident else myers_block_body(nw_block)
    score = np.full(n, qlen, dtype=np.int64)
    for blk in range(n_blocks):
        st = []
        for _ in range(nw_block):
            st += [np.full(n, FULL, np.uint32), np.zeros(n, np.uint32)]
        st += [c.copy() for c in carry[0]] + [np.zeros(n, np.uint32) for _ in range(3)]
        planes = planes_all[blk * nw_block * 3:(blk + 1) * nw_block * 3]
        for r, ch in enumerate(query):
            if r > 0 and r % 32 == 0:           # the stream's CARRY event
                j = r // 32
                carry[j - 1] = [st[2 * nw_block + 3 + i].copy() for i in range(3)]
                for i in range(3):
                    st[2 * nw_block + i] = carry[j][i].copy()
            c = code.get(int(ch), 0)
            if peq_resident:
                body.simulate(st, [peq[c, blk * nw_block + w] for w in range(nw_block)])
            else:
                body.simulate(st, [], cls=c, planes=planes)
        tail = qlen % 32
        last = [st[2 * nw_block + 3 + i] for i in range(3)]
        if tail:
            last = [x << np.uint32(32 - tail) for x in last]   # left-align a partial chunk
        carry[n_chunks - 1] = [x.copy() for x in last]
        for w in range(nw_block):
            rem = slen - 32 * (blk * nw_block + w)
            mask = np.uint32(0xFFFFFFFF if rem >= 32 else (0 if rem <= 0 else (1 << rem) - 1))
            score += np.bitwise_count(st[2 * w] & mask).astype(np.int64)
            score -= np.bitwise_count(st[2 * w + 1] & mask).astype(np.int64)
    return (-score).astype(np.int16)


def code_planes(peq: np.ndarray) -> list:
    """[5][nw][n] Peq -> class-independent code planes B[w*3+i] (A=0 C=1 G=2 T=3 N=4)."""
    nw = peq.shape[1]
    out = []
    for w in range(nw):
        out += [peq[1, w] | peq[3, w], peq[2, w] | peq[3, w], peq[4, w]]
    return out


def myers_init_state(nw: int, groups: int, lanes: int) -> list:
    st = []
    for _ in range(groups * nw):
        st += [np.full(lanes, 0xFFFFFFFF, dtype=np.uint32), np.zeros(lanes, dtype=np.uint32)]
    return st


def myers_score(state: list, nw: int, qlen: int, slen: int, group: int = 0) -> np.ndarray:
    score = np.full(state[0].shape, qlen, dtype=np.int64)
    for w in range(nw):
        rem = slen - 32 * w
        mask = np.uint32(0xFFFFFFFF if rem >= 32 else (0 if rem <= 0 else (1 << rem) - 1))
        score += np.bitwise_count(state[(group * nw + w) * 2] & mask).astype(np.int64)
        score -= np.bitwise_count(state[(group * nw + w) * 2 + 1] & mask).astype(np.int64)
    return (-score).astype(np.int16)


# =================================================================================================
# Banded Myers (reference banded/BGSA_CPU/align_core.c:19-67, one row of cpu_cal_D0 + cpu_cal_score)
# =================================================================================================

def banded_body() -> Body:
    """32-bit band word (k <= 15).  State: S0 = VP, S1 = VN, S2 = errors counted on the lowest
    diagonal since row k.  E0, E1 = the two words of the row's class match string (offset by k+1
    bits) that the window straddles; scalars: $sh = row mod 32, $mask = band mask, $one = 1.

    The window is one funnel shift of the match string by the row number — the reference shifts
    five windows and feeds each a bit, every row (align_core.c:35-62).  Two slow-class
    instructions remain (the funnel shift and D0 >> 1); everything else is fast class.
    """
    b = Body()
    b.ALIGNBIT("w", "E1", "E0", "$sh")
    b.BITOP3("x", "w", "$mask", "S1", lambda w, m, vn: (w & m) | vn)
    b.AND("t", "x", "S0")
    b.ADD("t", "t", "S0")
    b.BITOP3("d0", "t", "S0", "x", lambda t, vp, x: (t ^ vp) | x)
    b.BITOP3("hp", "d0", "S0", "S1", lambda d, vp, vn: ~(d | vp) | vn)
    b.AND("hn", "d0", "S0")
    b.LSHR1("x2", "d0")
    b.AND("S1", "x2", "hp")
    b.BITOP3("S0", "hp", "x2", "hn", lambda hp, x2, hn: ~(hp | x2) | hn)
    b.BITOP3("e", "d0", "$one", "$one", lambda d, one, _o: ~d & one)
    b.ADD("S2", "S2", "e")
    return b


def banded_body64_sh64(g: int = 0) -> Body:
    """banded_body64 with D0 >> 1 as ONE v_lshrrev_b64 on the fixed pair (P<2g+1> : P<2g>) = (D0 hi : D0 lo) in place of the funnel
    shift and the plain shift: 21 VALU per row; everything else — the error count included — as banded_body64."""
    b = Body()
    dl, dh = f"P{2 * g}", f"P{2 * g + 1}"
    ops = banded_body64().ops
    err_bit = next(op for op in ops if op.dst == "e")      # reads D0's low word: before the in-place shift
    for op in ops:
        if op is err_bit:
            continue
        if op.kind == "alignbit" and op.dst == "x2l":
            b.ops.append(Op(err_bit.kind, "e", (dl,) + err_bit.srcs[1:], err_bit.imm))
            b.SHR64(dh, dl)
            continue
        if op.kind == "lshr1" and op.dst == "x2h":
            continue
        ren = lambda r: {"dl": dl, "dh": dh, "x2l": dl, "x2h": dh}.get(r, r)
        b.ops.append(Op(op.kind, ren(op.dst), tuple(ren(x) for x in op.srcs), op.imm))
    return b


def banded_funnel_body(groups: int = 1, wide: bool = False, sh64: bool = False) -> Body:
    """The funnel-shift rows (banded_body: k <= 15, banded_body64: k <= 31) for one or two subject groups per wave that share
    the stream and its scalar work, interleaved instruction by instruction as banded_cut_body's: group g's state is
    S<n*g .. n*g + n - 1> (n = 3: VP, VN, errors; wide n = 5: VP lo/hi, VN lo/hi, errors), its match-string words
    E<m*g .. m*g + m - 1> (m = 2, wide 3).  groups = 1 is banded_body() / banded_body64() itself.
    sh64: banded_body64_sh64 (its fixed register pair is per group already)."""
    n_state, n_eq = (5, 3) if wide else (3, 2)
    per = []
    for g in range(groups):
        if sh64:
            assert wide
            base = banded_body64_sh64(g)
        else:
            base = banded_body64() if wide else banded_body()
        ops = []
        for op in base.ops:
            def rename(r, g=g):
                if r.startswith("S"):
                    return f"S{n_state * g + int(r[1:])}"
                if r.startswith("E"):
                    return f"E{n_eq * g + int(r[1:])}"
                if r.startswith(("$", "P")):
                    return r
                return f"{r}_{g}"
            ops.append(Op(op.kind, rename(op.dst) if op.dst else "", tuple(rename(x) for x in op.srcs), op.imm,
                          rename(op.dst2) if op.dst2 else ""))
        # units of the interleave: single instructions, and a carry chain (add_co followed by its addc links) as ONE
        # unit — two groups' chains must not interleave, VCC is one register
        units = []
        for op in ops:
            if op.kind == "addc":
                units[-1].append(op)
            else:
                units.append([op])
        per.append(units)
    out = Body()
    for units in zip(*per):
        for u in units:
            out.ops.extend(u)
    return out


def banded_cut_rows(k: int) -> int:
    """Rows between two cuts of the one-word window form (bgsa_common.h: banded_cut_rows): the band's 2k + 1 bits,
    offset by up to rows - 1 bits, must fit one 32-bit word.  0: no room worth it (k > 12) — the funnel-shift row."""
    return 16 if k <= 8 else (8 if k <= 12 else 0)


def banded_cut_body(groups: int = 1) -> Body:
    """The sliding row WITHOUT its funnel shift (k <= 12), for one or two subject groups per wave.

    scripts/ubench/gen_banded_mix.py on MI355X (profiles/r03_ubench_banded_mix.txt): ONE half-rate-class instruction
    in a row makes the whole row issue at ~4.2 cycles per instruction instead of ~2.2 — banded_body's row costs 52.8
    cycles of vector issue with its v_alignbit and 28.4 with a v_mov in its place — and v_alignbit is in that class
    with any kind of shift operand, while v_lshrrev_b32 is not (by an immediate, a VGPR, or — among other
    instructions — an SGPR).  So the window of a row must come out of ONE register: the kernel keeps, per class, the
    32 bits of the match string that start at the last multiple of banded_cut_rows(k) rows ("cut": one v_alignbit
    per class every 16 rows for k <= 8, every 8 for k <= 12, in an event) and the row shifts that word right by
    `row mod cut` — at most cut - 1, which leaves the band's 2k + 1 bits inside the word.  Every instruction of the
    row is then fast class.  E<g> = group g's cut word of the row's class; state S<3g..3g+2> = VP, VN, errors.
    groups = 2: two subject groups share the wave (and the stream's scalar work); their rows are interleaved
    instruction by instruction, two independent dependency chains."""
    per = []
    for g in range(groups):
        b = Body()
        t = lambda n, g=g: f"{n}{g}"
        S = lambda i, g=g: f"S{3 * g + i}"
        b.LSHRS(t("w"), f"E{g}", "$sh")
        b.BITOP3(t("x"), t("w"), "$mask", S(1), lambda w, m, vn: (w & m) | vn)
        b.AND(t("t"), t("x"), S(0))
        b.ADD(t("t"), t("t"), S(0))
        b.BITOP3(t("d"), t("t"), S(0), t("x"), lambda t_, vp, x: (t_ ^ vp) | x)
        b.BITOP3(t("hp"), t("d"), S(0), S(1), lambda d, vp, vn: ~(d | vp) | vn)
        b.AND(t("hn"), t("d"), S(0))
        b.LSHR1(t("x2"), t("d"))
        b.AND(S(1), t("x2"), t("hp"))
        b.BITOP3(S(0), t("hp"), t("x2"), t("hn"), lambda hp, x2, hn: ~(hp | x2) | hn)
        b.BITOP3(t("e"), t("d"), "$one", "$one", lambda d, one, _o: ~d & one)
        b.ADD(S(2), S(2), t("e"))
        per.append(b.ops)
    out = Body()
    for ops in zip(*per):
        out.ops.extend(ops)
    return out


def banded_body64() -> Body:
    """64-bit band (16 <= k <= 31) as register pairs.  State: S0/S1 = VP lo/hi, S2/S3 = VN lo/hi,
    S4 = errors since row k.  E0..E2 = three consecutive 32-bit words of the class match string;
    scalars $sh, $mask_lo, $mask_hi, $one.  22 VALU per row, four of them slow class."""
    b = Body()
    b.ALIGNBIT("wl", "E1", "E0", "$sh")
    b.ALIGNBIT("wh", "E2", "E1", "$sh")
    b.BITOP3("xl", "wl", "$mask_lo", "S2", lambda w, m, vn: (w & m) | vn)
    b.BITOP3("xh", "wh", "$mask_hi", "S3", lambda w, m, vn: (w & m) | vn)
    b.AND("tl", "xl", "S0")
    b.AND("th", "xh", "S1")
    b.ADD_CO("tl", "tl", "S0")
    b.ADDC("th", "th", "S1")
    b.BITOP3("dl", "tl", "S0", "xl", lambda t_, vp, x: (t_ ^ vp) | x)
    b.BITOP3("dh", "th", "S1", "xh", lambda t_, vp, x: (t_ ^ vp) | x)
    b.BITOP3("hpl", "dl", "S0", "S2", lambda d, vp, vn: ~(d | vp) | vn)
    b.BITOP3("hph", "dh", "S1", "S3", lambda d, vp, vn: ~(d | vp) | vn)
    b.AND("hnl", "dl", "S0")
    b.AND("hnh", "dh", "S1")
    b.ALIGNBIT("x2l", "dh", "dl", "$one")      # (D0 >> 1) low word
    b.LSHR1("x2h", "dh")
    b.AND("S2", "x2l", "hpl")
    b.AND("S3", "x2h", "hph")
    b.BITOP3("S0", "hpl", "x2l", "hnl", lambda hp, x2, hn: ~(hp | x2) | hn)
    b.BITOP3("S1", "hph", "x2h", "hnh", lambda hp, x2, hn: ~(hp | x2) | hn)
    b.BITOP3("e", "dl", "$one", "$one", lambda d, one, _o: ~d & one)
    b.ADD("S4", "S4", "e")
    return b


BANDED_PHASE_MIN = 8   # bgsa_common.h: kBandedPhaseMin


def banded_phase_rows(k: int) -> int:
    """Rows a 32-bit word can hold the band of threshold k IN PLACE (bgsa_common.h: banded_phase_rows): the band's
    2k + 1 bits start at bit 0, move up one bit per row, and the bit above them must still exist in the phase's last
    row.  0 = no room worth using (k > 11): those thresholds run the sliding form (banded_body)."""
    rows = 31 - 2 * k
    return rows if rows >= BANDED_PHASE_MIN else 0


def banded_phase_body() -> Body:
    """The band held in place (k <= 11; BGSA_BANDED_IMPL=p, measured and NOT the default): for banded_phase_rows(k) rows
    the match window of each class is fixed (E0 = the phase's window of the row's class, one funnel shift per class
    and PHASE instead of one per row) and the band moves up one bit per row inside the register, which is the
    classical recurrence — HP and HN move left through x + x, fast class — instead of the sliding one (D0 >> 1,
    half-rate class).  Every instruction of the row is fast class, and the row alone is ≈ 7 % faster than
    banded_body's; but every phase ends with a re-anchoring event (two shifts, a popcount, five funnel shifts — all
    half-rate — plus the event's scalar path) that costs as much as 2.6 rows, so at k = 8 (15 rows per phase) the
    loop is 8 % SLOWER than the sliding form, and still 4 % slower at k = 2 (27 rows per phase).  DESIGN.md §4.4.
    State: S0 = VP, S1 = VN (at the row's offset), S2 = errors counted before this phase (since row k), S3 = the band
    mask at the row's offset, S4 = its lowest bit, S5 = the phase's error bits: ~D0 on the lowest diagonal, one bit
    per row (errors so far = S2 + popcount(S5)).  Bits outside the band hold no information: below it the masked
    match word generates no carry, above it only the first bit is read and it is the band's carry-out whatever VP
    holds there."""
    b = Body()
    b.BITOP3("x", "E0", "S1", "S3", lambda w, vn, m: (w | vn) & m)
    b.AND("t", "x", "S0")
    b.ADD("t", "t", "S0")
    b.BITOP3("d0", "t", "S0", "x", lambda t, vp, x: (t ^ vp) | x)
    b.BITOP3("hp", "d0", "S0", "S1", lambda d, vp, vn: ~(d | vp) | vn)
    b.AND("hn", "d0", "S0")
    b.ADD("hp", "hp", "hp")
    b.ADD("hn", "hn", "hn")
    b.AND("S1", "d0", "hp")
    b.BITOP3("S0", "d0", "hp", "hn", lambda d, hp, hn: ~(d | hp) | hn)
    b.BITOP3("S5", "S5", "d0", "S4", lambda a, d, sel: a | (~d & sel))
    b.ADD("S3", "S3", "S3")
    b.ADD("S4", "S4", "S4")
    return b


def banded_last_check(length: int, k: int) -> int:
    """Rows after which the reference runs its last err > max_err test (banded.hip)."""
    h = k
    return length if length <= 64 else max(64, length - h)


BANDED_CHECK_ROWS = 8   # bgsa_common.h: kBandedCheckRows
BANDED_LATE_ROWS = 48   # bgsa_common.h: kBandedLateRows


def banded_tokens(length: int, k: int, word_bits: int = 32, phase: int = 0, cut: int = 0):
    """The per-query token sequence of the banded stream, query characters as ('row', r):
    ('event', bits) with bits 1 = reset the error count (row k), 2 = advance the match-string
    words (every word_bits rows), 4 = test err > limit on all lanes, 8 = latch the reject mask
    (the reference's last checkpoint), 16 = re-anchor the band (every `phase` rows; 0, the default: the sliding
    form, no such event), 32 = cut the next one-word window (every `cut` rows that do not also advance the words;
    0: the funnel-shift row, no such event).  Mirrors banded_stream_layout() in bgsa_common.h."""
    last = banded_last_check(length, k)
    out, pending = [], 0
    for r in range(length):
        if r == k:
            pending |= 1
        if r > 0 and r % word_bits == 0:
            pending |= 2
        if phase and r > 0 and r % phase == 0:
            pending |= 16
        if cut and r > 0 and r % cut == 0 and r % word_bits != 0:
            pending |= 32
        if pending:
            out.append(("event", pending))
            pending = 0
        out.append(("row", r))
        done = r + 1
        if done > k and done <= last:
            if done == last:
                pending |= 4 | 8
            elif done % BANDED_CHECK_ROWS == 0 and done - k > k + 1 and \
                    not (done > k + BANDED_LATE_ROWS and done % (2 * BANDED_CHECK_ROWS) != 0):
                pending |= 4
    if pending:
        out.append(("event", pending))
    return out


BANDED_SINGLE, BANDED_END, BANDED_REFILL, BANDED_EVENT = 25, 30, 31, 63


def banded_stream_bytes(length: int, k: int, codes, phase: int = 0, cut: int = 0) -> list:
    """The packed banded stream, byte for byte as bgsa_common.h: banded_stream_layout() writes it:
    banded_tokens() with two consecutive rows folded into one token (5*a + b) wherever no event sits
    between them, one-row tokens 25 + c otherwise, EVENT = {63, bits}, 7 payload bytes + REFILL per
    8-byte window, a two-byte token never split across windows, END padding and one spare END window."""
    out, slot = [], 0

    def put(b):
        nonlocal slot
        out.append(b)
        slot += 1
        if slot == 7:
            out.append(BANDED_REFILL)
            slot = 0

    def put_event(bits):
        nonlocal slot
        if slot == 6:
            out.extend([BANDED_REFILL, BANDED_END])
            slot = 0
        put(BANDED_EVENT)
        put(bits)

    toks = banded_tokens(length, k, phase=phase, cut=cut)
    i = 0
    while i < len(toks):
        kind, val = toks[i]
        if kind == "event":
            put_event(val)
            i += 1
        elif i + 1 < len(toks) and toks[i + 1][0] == "row":
            put(5 * int(codes[val]) + int(codes[toks[i + 1][1]]))
            i += 2
        else:
            put(BANDED_SINGLE + int(codes[val]))
            i += 1
    put(BANDED_END)
    while len(out) % 8:
        out.append(BANDED_END)
    out.extend([BANDED_END] * 8)
    return out


def _banded_simulate_cut(subjects: np.ndarray, query: np.ndarray, k: int, cut: int, groups: int) -> np.ndarray:
    """banded_cut_kernel<G> on the CPU: banded_cut_body(groups) — the subjects split into `groups` halves that share
    the token stream —, the cut / advance events as the generated loop runs them on its two registers per class (advance:
    A <- B, B <- the next word; cut: A <- {B >> bits cut so far, A} >> cut; the row shifts A by the rows since), the
    final band walk."""
    n, length = subjects.shape
    assert n % groups == 0
    code = np.zeros(256, dtype=np.uint8)
    for ch, c in zip(b"ACGTN", range(5)):
        code[ch] = c
    mapped = code[subjects]
    nwords = (length + 31) // 32 + 3
    mext = np.zeros((5, nwords, n), dtype=np.uint32)
    for p in range(length):
        i = p + k + 1
        for c in range(5):
            mext[c, i // 32] |= (mapped[:, p] == c).astype(np.uint32) << np.uint32(i % 32)
    h = k
    band_mask = np.uint32((1 << (2 * k + 1)) - 1)
    body = banded_cut_body(groups)
    per = n // groups
    sl = [slice(g * per, (g + 1) * per) for g in range(groups)]
    st = [np.zeros(per, dtype=np.uint32) for _ in range(3 * groups)]
    dead = np.zeros(n, dtype=bool)
    wi, sh, off = 0, 0, 0
    E = [[mext[c, 0, sl[g]].copy() for c in range(5)] for g in range(groups)]   # register A; row 0: the first word itself
    Bw = [[mext[c, 1, sl[g]].copy() for c in range(5)] for g in range(groups)]  # register B

    def alignbit(hi, lo, s):
        return ((((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> np.uint64(s)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)

    qcode = code[query]
    stopped = False
    for kind, val in banded_tokens(length, k, phase=0, cut=cut):
        if kind == "event":
            if val & 4:
                over = np.concatenate([st[3 * g + 2] > np.uint32(h + 1) for g in range(groups)])
                if val & 8:
                    dead = over.copy()
                if over.all():
                    stopped = True
                    break
            if val & 1:
                for g in range(groups):
                    st[3 * g + 2] = np.zeros(per, dtype=np.uint32)
            if val & 2:
                wi, sh, off = wi + 1, 0, 0
                for g in range(groups):
                    for c in range(5):
                        E[g][c] = Bw[g][c]
                        Bw[g][c] = mext[c, wi + 1, sl[g]].copy()
            if val & 32:
                for g in range(groups):
                    for c in range(5):
                        E[g][c] = alignbit(Bw[g][c] >> np.uint32(off), E[g][c], cut)
                off, sh = off + cut, 0
                for g in range(groups):      # what the window must be: bits off.. of the 64-bit pair of this advance
                    for c in range(5):
                        assert np.array_equal(E[g][c], alignbit(mext[c, wi + 1, sl[g]], mext[c, wi, sl[g]], off))
        else:
            c = int(qcode[val])
            assert sh < cut and sh + 2 * k + 1 <= 32
            body.simulate(st, [E[g][c] for g in range(groups)], scalars={"$sh": sh, "$mask": band_mask, "$one": 1})
            sh += 1
    if stopped:
        return np.full(n, 127, dtype=np.int8)
    out = np.empty(n, dtype=np.int8)
    for g in range(groups):
        err = st[3 * g + 2].astype(np.int64) + k
        best = err.copy()
        for i in range(h + 1):
            err = err + ((st[3 * g] >> np.uint32(i)) & 1) - ((st[3 * g + 1] >> np.uint32(i)) & 1)
            best = np.minimum(best, err)
        out[sl[g]] = np.where(dead[sl[g]], 127, best).astype(np.int8)
    return out


def banded_simulate(subjects: np.ndarray, query: np.ndarray, k: int, wide: bool | None = None,
                    phase: int = 0, cut: int = 0, groups: int = 1) -> np.ndarray:
    """Whole banded pipeline on the CPU with the shipped row body: Mext preprocess, token
    stream, events, final band walk.  Returns int8 results like the kernel.  wide = the 64-bit
    band body (register pairs), the default for k > 15."""
    if wide is None:
        wide = k > 15
    if wide:
        return _banded_simulate64(subjects, query, k)
    if phase:
        return _banded_simulate_phase(subjects, query, k, phase)
    if cut:
        return _banded_simulate_cut(subjects, query, k, cut, groups)
    n, length = subjects.shape
    code = np.zeros(256, dtype=np.uint8)
    for ch, c in zip(b"ACGTN", range(5)):
        code[ch] = c
    mapped = code[subjects]
    nwords = (length + 31) // 32 + 2
    mext = np.zeros((5, nwords, n), dtype=np.uint32)
    for p in range(length):
        i = p + k + 1
        for c in range(5):
            mext[c, i // 32] |= (mapped[:, p] == c).astype(np.uint32) << np.uint32(i % 32)
    h = k
    band_mask = np.uint32((1 << (2 * k + 1)) - 1)
    body = banded_body()
    st = [np.zeros(n, dtype=np.uint32) for _ in range(3)]
    dead = np.zeros(n, dtype=bool)
    wi, sh = 0, 0
    qcode = code[query]
    for kind, val in banded_tokens(length, k, phase=0):
        if kind == "event":
            if val & 4:
                over = st[2] > np.uint32(h + 1)
                if val & 8:
                    dead = over.copy()
                if over.all():
                    dead[:] = True
                    break
            if val & 1:
                st[2] = np.zeros(n, dtype=np.uint32)
            if val & 2:
                wi, sh = wi + 1, 0
        else:
            c = int(qcode[val])
            body.simulate(st, [mext[c, wi], mext[c, wi + 1]], scalars={"$sh": sh, "$mask": band_mask, "$one": 1})
            sh += 1
    err = st[2].astype(np.int64) + k
    best = err.copy()
    for i in range(h + 1):
        err = err + ((st[0] >> np.uint32(i)) & 1) - ((st[1] >> np.uint32(i)) & 1)
        best = np.minimum(best, err)
    return np.where(dead, 127, best).astype(np.int8)


def _popcount32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    return sum(((x >> np.uint64(i)) & np.uint64(1)) for i in range(32)).astype(np.uint32)


def _banded_simulate_phase(subjects: np.ndarray, query: np.ndarray, k: int, phase: int) -> np.ndarray:
    """banded_asm_kernel<false, true> on the CPU: banded_phase_body, the token stream with re-anchor events, the
    kernel's prologue (window = first match word, mask at offset 0) and epilogue (shift by the rows of the last
    phase, fold the collected error bits)."""
    n, length = subjects.shape
    code = np.zeros(256, dtype=np.uint8)
    for ch, c in zip(b"ACGTN", range(5)):
        code[ch] = c
    mapped = code[subjects]
    nwords = (length + 31) // 32 + 2
    mext = np.zeros((5, nwords, n), dtype=np.uint32)
    for p in range(length):
        i = p + k + 1
        for c in range(5):
            mext[c, i // 32] |= (mapped[:, p] == c).astype(np.uint32) << np.uint32(i % 32)
    h = k
    band_mask = np.uint32((1 << (2 * k + 1)) - 1)
    body = banded_phase_body()
    z = lambda: np.zeros(n, dtype=np.uint32)
    st = [z(), z(), z(), np.full(n, band_mask, np.uint32), np.ones(n, np.uint32), z()]
    W = [mext[c, 0].copy() for c in range(5)]
    dead = np.zeros(n, dtype=bool)
    wi, sh, in_phase = 0, 0, 0
    qcode = code[query]
    stopped = False
    for kind, val in banded_tokens(length, k, phase=phase):
        if kind == "event":
            if val & 4:
                over = (st[2] + _popcount32(st[5])) > np.uint32(h + 1)
                if val & 8:
                    dead = over.copy()
                if over.all():
                    dead[:] = True
                    stopped = True
                    break
            if val & 1:
                st[2], st[5] = z(), z()
            if val & 2:
                wi, sh = wi + 1, 0
            if val & 16:
                assert in_phase == phase
                st[0] = st[0] >> np.uint32(phase)
                st[1] = st[1] >> np.uint32(phase)
                st[2] = st[2] + _popcount32(st[5])
                st[3], st[4], st[5] = np.full(n, band_mask, np.uint32), np.ones(n, np.uint32), z()
                for c in range(5):
                    pair = (mext[c, wi + 1].astype(np.uint64) << np.uint64(32)) | mext[c, wi].astype(np.uint64)
                    W[c] = ((pair >> np.uint64(sh)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                in_phase = 0
        else:
            c = int(qcode[val])
            body.simulate(st, [W[c]])
            sh += 1
            in_phase += 1
    if stopped:
        return np.full(n, 127, dtype=np.int8)
    final_shift = length - phase * ((length - 1) // phase)
    assert final_shift == in_phase
    vp = st[0] >> np.uint32(final_shift)
    vn = st[1] >> np.uint32(final_shift)
    err = (st[2] + _popcount32(st[5])).astype(np.int64) + k
    best = err.copy()
    for i in range(h + 1):
        err = err + ((vp >> np.uint32(i)) & 1) - ((vn >> np.uint32(i)) & 1)
        best = np.minimum(best, err)
    return np.where(dead, 127, best).astype(np.int8)


def _banded_simulate64(subjects: np.ndarray, query: np.ndarray, k: int) -> np.ndarray:
    n, length = subjects.shape
    code = np.zeros(256, dtype=np.uint8)
    for ch, c in zip(b"ACGTN", range(5)):
        code[ch] = c
    mapped = code[subjects]
    nwords = (length + 31) // 32 + 3
    mext = np.zeros((5, nwords, n), dtype=np.uint32)
    for p in range(length):
        i = p + k + 1
        for c in range(5):
            mext[c, i // 32] |= (mapped[:, p] == c).astype(np.uint32) << np.uint32(i % 32)
    h = k
    band = (1 << (2 * k + 1)) - 1
    body = banded_body64()
    st = [np.zeros(n, dtype=np.uint32) for _ in range(5)]
    dead = np.zeros(n, dtype=bool)
    wi, sh = 0, 0
    qcode = code[query]
    for kind, val in banded_tokens(length, k, phase=0):
        if kind == "event":
            if val & 4:
                over = st[4] > np.uint32(h + 1)
                if val & 8:
                    dead = over.copy()
                if over.all():
                    dead[:] = True
                    break
            if val & 1:
                st[4] = np.zeros(n, dtype=np.uint32)
            if val & 2:
                wi, sh = wi + 1, 0
        else:
            c = int(qcode[val])
            body.simulate(st, [mext[c, wi], mext[c, wi + 1], mext[c, wi + 2]],
                          scalars={"$sh": sh, "$mask_lo": band & 0xFFFFFFFF, "$mask_hi": band >> 32, "$one": 1})
            sh += 1
    vp = st[0].astype(np.uint64) | (st[1].astype(np.uint64) << np.uint64(32))
    vn = st[2].astype(np.uint64) | (st[3].astype(np.uint64) << np.uint64(32))
    err = st[4].astype(np.int64) + k
    best = err.copy()
    for i in range(h + 1):
        err = err + ((vp >> np.uint64(i)) & np.uint64(1)).astype(np.int64) - ((vn >> np.uint64(i)) & np.uint64(1)).astype(np.int64)
        best = np.minimum(best, err)
    return np.where(dead, 127, best).astype(np.int8)


# =================================================================================================
# BitPAl packed for any integer scores (match M > mismatch I >= 2*gap G, G < 0): the scheme the
# reference's generator emits per score set (generator/.../BitPAlGenerator.java:151-534 packed form,
# ScoreMsg.java:23-31 for the value ranges), restated on normalised differences.
#
#   u_j = dH(row above, column j) - G   in [0, C],  C = M - 2G      (stored unsigned on bits(C) planes)
#   v_j = dV(column j)            - G   in [0, C]
#   v_j = max(0, w_j - u_j),  w_j = C at a match, else max(D, v_{j-1}),  D = I - 2G
#   new u_j = max(0, max(W_j, u_j) - v_{j-1}),  W_j = C at a match, D otherwise
#
# Only the K = M - I values above D survive a mismatch unchanged, so the row resolves the chain
# v_{j-1} -> v_j per value class C, C-1, ..., D+1 with one add-with-carry run propagation each (the
# value is kept across columns with u = 0) and treats everything else as class D.
# =================================================================================================

@dataclass(frozen=True)
class BitpalScores:
    match: int = 2
    mismatch: int = -3
    gap: int = -5

    def __post_init__(self):
        if not (self.match > self.mismatch and self.mismatch >= 2 * self.gap and self.gap < 0):
            raise ValueError("BitPAl needs match > mismatch >= 2*gap and gap < 0")

    @property
    def C(self) -> int: return self.match - 2 * self.gap          # largest normalised difference
    @property
    def D(self) -> int: return self.mismatch - 2 * self.gap       # the mismatch class
    @property
    def K(self) -> int: return self.match - self.mismatch         # value classes above D
    @property
    def nb(self) -> int: return self.C.bit_length()               # magnitude planes
    @property
    def planes(self) -> int: return self.nb                       # unsigned u: no sign plane
    @property
    def chains(self) -> int: return 1 + (1 if BITPAL_ONE_CHAIN else 2) * (self.K - 1) + self.nb
    @property
    def tag(self) -> str:
        f = lambda v: f"m{-v}" if v < 0 else f"{v}"
        return f"{f(self.match)}_{f(self.mismatch)}_{f(self.gap)}"

    def weights(self) -> tuple:
        """Per-plane weight of a set bit in the final score: u = sum 2^i p_i."""
        return tuple(1 << i for i in range(self.nb))


BITPAL_DEFAULT = BitpalScores(2, -3, -5)
BITPAL_SCHEDULE_WINDOW = int(__import__('os').environ.get('BGSA_GEN_BITPAL_WINDOW', '48'))   # 0 = program order (A/B)
# build-time A/B switches of round 4's two rewrites of the row body (see bitpal_scores_body; scripts/r04_bitpal_ab.sh,
# profiles/r04_bitpal_ab.txt, 10k x 1M x 150 bp on one box): reading "u <= D" off its plane — one instruction fewer — measured
# 6,893 -> 6,854 ms and is on; one addition chain per class instead of two — same count, 13 -> 9 chains — measured 6,893 -> 6,949 ms
# (and the chip held a lower clock under it) and is OFF
BITPAL_ONE_CHAIN = __import__('os').environ.get('BGSA_GEN_BITPAL_ONE_CHAIN', '0') != '0'
BITPAL_INLINE_LE = __import__('os').environ.get('BGSA_GEN_BITPAL_INLINE_LE', '1') != '0'
# round 4, second step: new u = max(w, u) - v_in needs no clamp (see bitpal_scores_body: "the cell identity"); '0' builds the
# previous form, new u = max(0, max(W, u) - v_in), for A/B
BITPAL_CELL_MAX = __import__('os').environ.get('BGSA_GEN_BITPAL_CELL_MAX', '1') != '0'
# the one-hot mask of u = K - 1 has ONE reader — the lowest class's seed term with the top class — so that product is formed from the
# planes and the class mask directly: one instruction per word fewer ('0': the mask is built like the others, for A/B)
BITPAL_FOLD_LAST_Z = __import__('os').environ.get('BGSA_GEN_BITPAL_FOLD_LAST_Z', '1') != '0'
# the run mask (u = 0 at a MISMATCH) is not needed: the classes may propagate through every u = 0 column — what runs through a u = 0
# match column is covered, column by column, by the top class's own propagation from that very column, and the extraction masks
# with the top class instead of the match mask.  One instruction per word fewer ('0': the run mask, for A/B).
BITPAL_NO_RUN = __import__('os').environ.get('BGSA_GEN_BITPAL_NO_RUN', '1') != '0'



class _Bool:
    """Boolean-expression helper over a Body with common-subexpression sharing per word."""

    def __init__(self, body: Body, prefix: str):
        self.b = body
        self.memo = {}
        self.n = 0
        self.prefix = prefix

    def _name(self, hint):
        self.n += 1
        return f"{hint}{self.prefix}_{self.n}"

    def op3(self, fn, a, b_, c, hint="x"):
        # canonical operand order, so that e.g. OR3(p0, p1, p2) and OR3(p2, p1, p0) are one instruction
        order = sorted(range(3), key=lambda i: (a, b_, c)[i])
        ops = [(a, b_, c)[i] for i in order]

        def g(x0, x1, x2, order=order, fn=fn):
            args = [None, None, None]
            for pos, i in enumerate(order):
                args[i] = (x0, x1, x2)[pos]
            return fn(*args)

        key = (tt(g), *ops)
        if key not in self.memo:
            self.memo[key] = self.b.BITOP3(self._name(hint), ops[0], ops[1], ops[2], g)
        return self.memo[key]

    def op2(self, kind, a, b_, hint="x"):
        key = (kind, a, b_)
        if key not in self.memo:
            self.memo[key] = getattr(self.b, kind)(self._name(hint), a, b_)
        return self.memo[key]

    def or_all(self, names, hint="o"):
        names = list(names)
        while len(names) > 1:
            if len(names) == 2:
                return self.op2("OR", names[0], names[1], hint)
            merged = self.op3(lambda a, b_, c: a | b_ | c, names[0], names[1], names[2], hint)
            names = [merged] + names[3:]
        return names[0]

    def and_pattern(self, planes, bits, hint="z"):
        """AND over planes of (plane if bit else ~plane); planes given MSB first.  Chunks are taken
        from the MSB side so that patterns sharing their high bits share the partial products."""
        n = len(planes)
        lit = lambda x, bit: x if bit else ~x
        first = 3 if n >= 3 and (n - 3) % 2 == 0 else 2
        if first == 3:
            b0, b1, b2 = bits[:3]
            cur = self.op3(lambda a, b_, c: lit(a, b0) & lit(b_, b1) & lit(c, b2), planes[0], planes[1], planes[2], hint)
        else:
            b0, b1 = bits[:2]
            cur = self.op3(lambda a, b_, c: lit(a, b0) & lit(b_, b1), planes[0], planes[1], planes[1], hint)
        i = first
        while i < n:
            b0, b1 = bits[i], bits[i + 1]
            cur = self.op3(lambda a, b_, c: a & lit(b_, b0) & lit(c, b1), cur, planes[i], planes[i + 1], hint)
            i += 2
        return cur


@__import__("functools").lru_cache(maxsize=256)
def bitpal_scores_body(nw: int, sc: BitpalScores = BITPAL_DEFAULT) -> Body:
    """(Memoised: callers treat the returned Body as read-only.)  The row body for `nw` words, with the single-use one-hot mask folded into
    its seed product where that saves an instruction for this score set (it does when the mask shares no partial product with the others:
    2/-3/-5 64 -> 63; 10/-9/-15 would lose one)."""
    if BITPAL_FOLD_LAST_Z and _bitpal_scores_body(1, sc, True).valu_count() < _bitpal_scores_body(1, sc, False).valu_count():
        return _bitpal_scores_body(nw, sc, True)
    return _bitpal_scores_body(nw, sc, False)


@__import__("functools").lru_cache(maxsize=512)
def _bitpal_scores_body(nw: int, sc: BitpalScores, fold_last_z: bool) -> Body:
    """Row body for `nw` words.  State S[w*B + i] = plane i (weight 2^i) of the UNSIGNED value u of word
    w's 32 columns, B = sc.planes = bits(C) (the reference keeps the two's complement of -u in one more
    plane, align_core.c:191-214; the unsigned form saves that plane and four instructions per word).
    E[w] = match mask.  Chains (in order): the top-class run, then per lower class the seed shift and its run (one chain
    for both under BGSA_GEN_BITPAL_ONE_CHAIN=1, measured slower), then the B plane shifts — sc.chains in all."""
    B, C, D, K = sc.planes, sc.C, sc.D, sc.K
    b = Body()
    U = lambda w, i: f"S{w * B + i}"
    E = lambda w: f"E{w}"
    t = lambda name, w: f"{name}_{w}"
    W = range(nw)
    z = {}      # (value of u, word) -> one-hot mask name
    anym = {}   # word -> mismatch columns whose u <= D
    le_plane = {}   # word -> the ONE plane p with "u <= D  <=>  ~p" (D + 1 a power of two): no mask of its own is built
    dv = {}     # (class offset c, word) -> columns whose incoming v is C - c (c = 0: or a match)
    prop, stop = {}, {}   # word -> the columns a class value runs through unchanged; the columns that end up in the top class whatever arrives

    # ---- decode the u classes the seeds need, the "u <= D" mask, and the top-class run ----------
    for w in W:
        bx = _Bool(b, f"d{w}")
        planes = [U(w, i) for i in range(B)]
        msb_first = planes[::-1]
        if B == 1:
            z0 = bx.op3(lambda a, _a, __a: ~a, planes[0], planes[0], planes[0], "z0")
        elif B == 2:
            z0 = bx.op3(lambda a, b_, _b: ~(a | b_), planes[0], planes[1], planes[1], "z0")
        elif K > 1 and B > 3:
            # u == 0 shares the "high planes are zero" product with the small-u masks below
            z0 = bx.and_pattern(msb_first, [0] * B, "z")
        else:
            low = bx.or_all(planes[:B - 2], "lo") if B > 3 else planes[0]
            z0 = bx.op3(lambda a, b_, c: ~(a | b_ | c), low, planes[B - 2], planes[B - 1], "z0")   # u == 0
        for x in range(1, K):
            if fold_last_z and x == K - 1 and K > 2 and B > 1:
                continue      # read once, by seed K-1's first term: folded into that product below
            z[x, w] = bx.and_pattern(msb_first, [(x >> i) & 1 for i in range(B - 1, -1, -1)]) if B > 1 else planes[0]
        # u <= D: constant comparator from the LSB up (r = "the bits seen so far are <= the constant's";
        # a one bit of D ORs the complemented plane in, a zero bit ANDs it)
        step = lambda rv, p, bit: (~p | rv) if bit else (~p & rv)
        i = 0
        while i < B and (D >> i) & 1:      # trailing one bits of the constant: always satisfied
            i += 1
        if BITPAL_CELL_MAX:
            pass                               # max_planes() takes max(w, u) from the first subtract's borrow: nobody reads "u <= D"
        elif i >= B:
            anym[w] = bx.op3(lambda e, _e, __e: ~e, E(w), E(w), E(w), "any")          # every u is <= D
        else:
            r = None
            first = i
            i += 1
            # the first zero bit: r = ~plane
            pending = [("not", planes[first])]
            cur = None
            while i < B:
                b0 = (D >> i) & 1
                if i + 1 < B:
                    b1 = (D >> (i + 1)) & 1
                    if cur is None:
                        cur = bx.op3(lambda x1, x0, p, b0=b0, b1=b1: step(step(~p, x0, b0), x1, b1),
                                     planes[i + 1], planes[i], planes[first], "le")
                    else:
                        cur = bx.op3(lambda x1, x0, rv, b0=b0, b1=b1: step(step(rv, x0, b0), x1, b1),
                                     planes[i + 1], planes[i], cur, "le")
                    i += 2
                else:
                    if cur is None:
                        cur = bx.op3(lambda x0, p, _p, b0=b0: step(~p, x0, b0), planes[i], planes[first], planes[first], "le")
                    else:
                        cur = bx.op3(lambda x0, rv, _r, b0=b0: step(rv, x0, b0), planes[i], cur, cur, "le")
                    i += 1
            if cur is None and BITPAL_INLINE_LE:   # the comparison is just "~plane[first]": max_planes() reads that plane itself
                le_plane[w] = planes[first]
            elif cur is None:
                anym[w] = bx.op3(lambda p, e, _e: ~p & ~e, planes[first], E(w), E(w), "any")
            else:
                anym[w] = bx.op3(lambda le, e, _e: le & ~e, cur, E(w), E(w), "any")
        b.AND(t("seed", w), z0, E(w))
        if BITPAL_NO_RUN:
            prop[w] = z0
            (b.ADD_CO if w == 0 else b.ADDC)(t("sum", w), t("seed", w), z0)
            dv[0, w] = b.BITOP3(t("dvtop", w), t("sum", w), z0, E(w), lambda s, z_, e: (s ^ (z_ & ~e)) | e)
            stop[w] = dv[0, w]
            continue
        b.XOR(t("run", w), z0, t("seed", w))                       # u == 0 at a mismatch
        prop[w], stop[w] = t("run", w), E(w)
        (b.ADD_CO if w == 0 else b.ADDC)(t("sum", w), t("seed", w), z0)
        dv[0, w] = b.BITOP3(t("dvtop", w), t("sum", w), t("run", w), E(w), lambda s, r, e: (s ^ r) | e)

    def shifted_run(seed_name, c):
        # The seeds act one column up and spread over the run behind them: (seed << 1) + run.  A seed sits on a column with
        # u > 0, a run column has u = 0, so the two are disjoint and (seed << 1) + run = seed + (seed | run): ONE addition
        # with carry across the words instead of a shift chain and an addition chain (13 -> 9 chains for 2/-3/-5, same
        # instruction count).  Built, bit-exact, measured 0.8 % SLOWER on the full config 4: the A/B alternative.
        if not BITPAL_ONE_CHAIN:      # round 3's form: a shift chain, then an addition chain
            for w in W:
                (b.ADD_CO if w == 0 else b.ADDC)(t(seed_name, w), t(seed_name, w), t(seed_name, w))
            for w in W:
                (b.ADD_CO if w == 0 else b.ADDC)(t(f"rs{c}", w), t(seed_name, w), prop[w])
                dv[c, w] = b.BITOP3(t(f"dv{c}", w), t(f"rs{c}", w), prop[w], stop[w], lambda s, r, e: (s ^ r) & ~e)
            return
        for w in W:
            b.OR(t(f"sr{c}", w), t(seed_name, w), prop[w])
        for w in W:
            (b.ADD_CO if w == 0 else b.ADDC)(t(f"rs{c}", w), t(f"sr{c}", w), t(seed_name, w))
            dv[c, w] = b.BITOP3(t(f"dv{c}", w), t(f"rs{c}", w), prop[w], stop[w], lambda s, r, e: (s ^ r) & ~e)

    # ---- classes C-1 .. D+1: value C-c appears where an incoming class C-x meets u = c-x ----------
    for c in range(1, K):
        for w in W:
            if (c, w) not in z:      # fold_last_z: [u = K - 1] AND the top class in one product over the planes and the mask
                msb_first = [U(w, i) for i in range(B)][::-1]
                _Bool(b, f"q{w}").and_pattern(msb_first + [dv[0, w]], [(c >> i) & 1 for i in range(B - 1, -1, -1)] + [1], "zs")
                b.ops[-1].dst = t(f"seed{c}", w)      # the product's last instruction IS the seed's first term: name it so
            else:
                b.AND(t(f"seed{c}", w), z[c, w], dv[0, w])
            for x in range(1, c):
                b.BITOP3(t(f"seed{c}", w), z[c - x, w], dv[x, w], t(f"seed{c}", w), lambda a, b_, acc: (a & b_) | acc)
        shifted_run(f"seed{c}", c)

    # ---- w planes, v = max(0, w - u) = (w + ~u + 1) where that does not borrow, else 0 ------------------
    lit = {"reg": lambda x: x, "not": lambda x: ~x, "zero": lambda x: 0, "one": lambda x: 0xFF}

    def subtract(w, minuend, sub_name, out_name, top_out=None, clamp=True, out_reg=None):
        """out = max(0, minuend - sub) over B planes; minuend[i] = (kind, register name or None).  The top plane
        comes out clamped (into `top_out` if given); the caller ANDs planes 0..B-2 with the returned mask.
        clamp = False: the caller knows minuend >= sub — plain difference, plane i into out_reg(i), no borrow out of the top plane."""
        carry = None
        dst = (lambda i: out_reg(i)) if out_reg is not None else (lambda i: t(f"{out_name}{i}", w))
        for i in range(B):
            kind, name = minuend[i]
            f = lit[kind]
            src = name if name is not None else sub_name(i)
            if carry is None:      # bit 0: the +1 of the two's complement is the carry-in
                if kind == "reg":
                    b.XOR(dst(i), src, sub_name(i))
                else:
                    b.BITOP3(dst(i), src, sub_name(i), sub_name(i), lambda m, s, _s, f=f: f(m) ^ s)
                if B > 1 or clamp:
                    carry = b.BITOP3(t(f"{out_name}c", w), src, sub_name(i), sub_name(i), lambda m, s, _s, f=f: f(m) | ~s)
            elif i == B - 1 and not clamp:
                b.BITOP3(dst(i), src, sub_name(i), carry, lambda m, s, cy, f=f: f(m) ^ ~s ^ cy)
                carry = None
            elif i == B - 1:
                # top plane: its clamp (difference bit AND "no borrow out of this plane") is a function of the same three
                # inputs, so it costs no instruction of its own
                maj = lambda m, s, cy, f=f: (f(m) & ~s) | (f(m) & cy) | (~s & cy)
                prev = carry
                carry = b.BITOP3(t(f"{out_name}c{i}", w), src, sub_name(i), prev, maj)
                b.BITOP3(top_out or t(f"{out_name}{i}", w), src, sub_name(i), prev,
                         lambda m, s, cy, f=f, maj=maj: (f(m) ^ ~s ^ cy) & maj(m, s, cy))
            else:
                b.BITOP3(dst(i), src, sub_name(i), carry, lambda m, s, cy, f=f: f(m) ^ ~s ^ cy)
                carry = b.BITOP3(t(f"{out_name}c", w), src, sub_name(i), carry,
                                 lambda m, s, cy, f=f: (f(m) & ~s) | (f(m) & cy) | (~s & cy))
        return carry     # 1 = no borrow: the difference is >= 0; planes 0..B-2 still need their AND with it

    w_of, ok_of = {}, {}    # word -> the planes of w as subtract() takes them; the "w >= u" mask of the first subtract
    for w in W:
        bx = _Bool(b, f"w{w}")
        cls_mask = {C - c: dv[c, w] for c in range(K)}
        # plane i of w over the columns: classes above D contribute their masks, class D is "the rest".
        # Either spell every plane from the class masks alone, or spend the ops for rest = ~OR(all masks)
        # once and use it as one more mask — whichever costs fewer instructions overall.
        or_cost = lambda n: 0 if n <= 1 else n // 2
        have = [[v for v in cls_mask if (v >> i) & 1] for i in range(B)]
        lack = [[v for v in cls_mask if not (v >> i) & 1] for i in range(B)]
        d_bit = [(D >> i) & 1 for i in range(B)]

        def plan(with_rest):
            total, choice = (or_cost(K) if with_rest else 0), []
            for i in range(B):
                opts = []
                if with_rest:   # rest joins the side class D is on
                    opts.append((or_cost(len(have[i]) + d_bit[i]), "reg", have[i], bool(d_bit[i])))
                    opts.append((or_cost(len(lack[i]) + 1 - d_bit[i]), "not", lack[i], not d_bit[i]))
                elif d_bit[i]:
                    opts.append((or_cost(len(lack[i])), "not", lack[i], False))
                else:
                    opts.append((or_cost(len(have[i])), "reg", have[i], False))
                best = min(opts, key=lambda o: o[0])
                total += best[0]
                choice.append(best)
            return total, choice

        (cost_r, choice_r), (cost_n, choice_n) = plan(True), plan(False)
        use_rest = cost_r < cost_n
        rest = None
        if use_rest:
            allm = [cls_mask[v] for v in cls_mask]
            if len(allm) == 1:
                rest = bx.op3(lambda a, _a, __a: ~a, allm[0], allm[0], allm[0], "rest")
            elif len(allm) == 2:
                rest = bx.op3(lambda a, b_, _b: ~(a | b_), allm[0], allm[1], allm[1], "rest")
            else:
                head = bx.or_all(allm[:-2], "hi") if len(allm) > 3 else allm[0]
                rest = bx.op3(lambda a, b_, c: ~(a | b_ | c), head, allm[-2], allm[-1], "rest")
        wplanes = []   # (kind, name): plane i of w is name / ~name / 0 / 1
        for _cost, kind, vals, plus_rest in (choice_r if use_rest else choice_n):
            masks = [cls_mask[v] for v in vals] + ([rest] if plus_rest else [])
            if not masks:
                wplanes.append(("zero" if kind == "reg" else "one", None))
            else:
                wplanes.append((kind, bx.or_all(masks, "wp")))
        ok = subtract(w, wplanes, lambda i, w=w: U(w, i), "s")
        w_of[w], ok_of[w] = wplanes, ok
        for i in range(B - 1 if B > 1 else B):
            b.AND(t(f"s{i}", w), t(f"s{i}", w), ok)

    # ---- max(W, u): C at a match, D at a mismatch with u <= D, else u -------------------------------
    def max_planes(w):
        if BITPAL_CELL_MAX:
            # The cell identity: with a = u, b = v_in and W the cell's substitution value, v = max(0, max(W, b) - a) and
            # new u = max(0, max(W, a) - b) are m - a and m - b for m = max(W, a, b) — the largest of the three can be clamped by
            # neither.  w = max(W, b) is in hand (the class masks), "w >= u" is the first subtract's borrow: m = w or u by that
            # mask, one instruction per plane (what max(W, u) cost), and the second subtract needs no clamp and no borrow out of
            # its top plane: four instructions per word fewer (2/-3/-5: 68 -> 64), and the "u <= D" comparator has no reader left.
            for i in range(B):
                kind, name = w_of[w][i]
                f = lit[kind]
                if name is None:
                    b.BITOP3(t(f"g{i}", w), U(w, i), ok_of[w], ok_of[w], lambda u, ok, _ok, f=f: (f(0) & ok) | (u & ~ok))
                else:
                    b.BITOP3(t(f"g{i}", w), name, U(w, i), ok_of[w], lambda wp, u, ok, f=f: (f(wp) & ok) | (u & ~ok))
            return
        for i in range(B):
            cb, db = 0xFF * ((C >> i) & 1), 0xFF * ((D >> i) & 1)
            if w in le_plane:      # "u <= D" is the complement of one plane: read it directly (one instruction per word saved)
                p = le_plane[w]
                if U(w, i) == p:
                    b.BITOP3(t(f"g{i}", w), p, p, E(w), lambda u, _u, e, cb=cb, db=db: (cb & e) | (db & ~u & ~e) | (u & ~e))
                else:
                    b.BITOP3(t(f"g{i}", w), U(w, i), p, E(w),
                             lambda u, p_, e, cb=cb, db=db: (cb & e) | (db & ~p_ & ~e) | (u & p_ & ~e))
                continue
            b.BITOP3(t(f"g{i}", w), U(w, i), anym[w], E(w),
                     lambda u, a, e, cb=cb, db=db: (cb & e) | (db & a & ~e) | (u & ~a & ~e))

    # ---- v one column up: B shift chains, the max-plane work interleaved into the first ---------------
    for i in range(B):
        for w in W:
            (b.ADD_CO if w == 0 else b.ADDC)(t(f"s{i}", w), t(f"s{i}", w), t(f"s{i}", w))
            if i == 0:
                max_planes(w)

    # ---- new u = max(0, max(W, u) - v_in) -------------------------------------------------------------
    for w in W:
        if BITPAL_CELL_MAX:
            subtract(w, [("reg", t(f"g{i}", w)) for i in range(B)], lambda i, w=w: t(f"s{i}", w), "r", clamp=False,
                     out_reg=lambda i, w=w: U(w, i))
            continue
        ok = subtract(w, [("reg", t(f"g{i}", w)) for i in range(B)], lambda i, w=w: t(f"s{i}", w), "r",
                      top_out=U(w, B - 1) if B > 1 else None)
        for i in range(B - 1 if B > 1 else B):
            b.AND(U(w, i), t(f"r{i}", w), ok)
    # chains first, their wait states filled with independent work (no extra registers)
    return schedule(b, BITPAL_SCHEDULE_WINDOW) if BITPAL_SCHEDULE_WINDOW else b


def bitpal_scores_init_state(nw: int, lanes: int, sc: BitpalScores = BITPAL_DEFAULT, semi: bool = False) -> list:
    """Row 0: every column at dH = G (global), or at dH = 0, i.e. u = -G (semi-global: the
    generator's writeBitInitStr, BitPAlGenerator.java:2201-2218)."""
    stored = -sc.gap if semi else 0                                   # u itself
    return [np.full(lanes, 0xFFFFFFFF if (stored >> i) & 1 else 0, dtype=np.uint32)
            for _ in range(nw) for i in range(sc.planes)]


def bitpal_column_values(state: list, w: int, sc: BitpalScores) -> np.ndarray:
    """u of the 32 columns of word w: [lanes, 32] int64."""
    B = sc.planes
    bits = np.arange(32, dtype=np.uint32)
    u = np.zeros((state[0].shape[0], 32), dtype=np.int64)
    for i, wt in enumerate(sc.weights()):
        u += wt * ((state[w * B + i][:, None] >> bits[None, :]) & np.uint32(1)).astype(np.int64)
    return u


def bitpal_scores_score(state: list, nw: int, qlen: int, slen: int, sc: BitpalScores = BITPAL_DEFAULT,
                        semi: bool = False) -> np.ndarray:
    """Global: S[m][n] = G*(m + n) + sum over subject columns of u.  Semi-global: the maximum of
    S[m][j] = G*m + sum_{c <= j} (u_c + G) over j = 0..n (genPackedScore, BitPAlGenerator.java:78-116)."""
    B = sc.planes
    if semi:
        u = np.concatenate([bitpal_column_values(state, w, sc) for w in range(nw)], axis=1)[:, :slen]
        run = sc.gap * qlen + np.cumsum(u + sc.gap, axis=1)
        best = np.maximum(run.max(axis=1, initial=-(1 << 40)), sc.gap * qlen)
        return best.astype(np.int16)
    score = np.full(state[0].shape, sc.gap * (qlen + slen), dtype=np.int64)
    for w in range(nw):
        rem = slen - 32 * w
        mask = np.uint32(0xFFFFFFFF if rem >= 32 else (0 if rem <= 0 else (1 << rem) - 1))
        for i, wt in enumerate(sc.weights()):
            score += wt * np.bitwise_count(state[w * B + i] & mask).astype(np.int64)
    return score.astype(np.int16)


def make_blocked(body: Body, n_state: int, n_slots: int | None = None):
    """Column-block form of a row body: every carry chain enters and leaves through a carry word
    (32 rows per word, first row in bit 31).  Chain k reads its carry-in with `x + x` on state
    register S[n_state + k] (VCC = next row's bit) and appends its carry-out with `x + x + vcc` on
    S[n_state + n_chains + k].  Returns (new body, initial carry-in bit per chain): a chain that
    started with SETC1 has carry-in 1 in the first block, the others 0.
    (myers_block_body is this transformation of myers_planes_body, written out by hand.)"""
    chains = []          # (index of the chain's first op, initial carry)
    for i, op in enumerate(body.ops):
        if op.kind == "add_co":
            chains.append((i, 0))
        elif op.kind == "setc1":
            chains.append((i, 1))
    n = len(chains) if n_slots is None else n_slots      # n_slots: the kernel's carry-word pairs when it has more than the body has chains
    assert len(chains) <= n
    starts = {i: k for k, (i, _) in enumerate(chains)}
    out = Body()
    open_chain = None
    for i, op in enumerate(body.ops):
        if i in starts:
            if open_chain is not None:
                out.ADDC(f"S{n_state + n + open_chain}", f"S{n_state + n + open_chain}", f"S{n_state + n + open_chain}")
            k = starts[i]
            cin = f"S{n_state + k}"
            out.ADD_CO(cin, cin, cin)
            open_chain = k
            if op.kind == "add_co":
                out.ops.append(Op("addc", op.dst, op.srcs, 0))
            continue  # setc1 itself is dropped: the carry-in word supplies the 1
        out.ops.append(op)
    if open_chain is not None:
        out.ADDC(f"S{n_state + n + open_chain}", f"S{n_state + n + open_chain}", f"S{n_state + n + open_chain}")
    return out, [c for _, c in chains]


def make_blocked_packed(body: Body, n_state: int):
    """Column-block form for bodies with MANY carry chains: the carries of one row travel as bits of a few words —
    chain k in word k // 32, first chain in bit 31 — instead of one word per chain and 32 rows (make_blocked), which
    costs two registers per chain: 86 for the 43 chains of 10/-9/-15, more than a column-block kernel can hold.
    Here the block holds ceil(chains / 32) carry-in words and as many carry-out words whatever the score set:
    chain k takes its carry-in with `x + x` on S[n_state + k // 32] (VCC = the word's top bit) and appends its
    carry-out with `x + x + vcc` on S[n_state + W + k // 32]; a last, partly filled word is moved up at the end of the
    row so that its first chain sits in bit 31 again.  The words are exchanged with the neighbouring blocks EVERY
    row (the row loop loads the next row's words while it computes this one's and stores its own behind the body):
    two or three loads and stores per row against hundreds of vector instructions.  Returns (body, initial carry-in
    bit per chain, words)."""
    chains = []
    for i, op in enumerate(body.ops):
        if op.kind == "add_co":
            chains.append((i, 0))
        elif op.kind == "setc1":
            chains.append((i, 1))
    n = len(chains)
    n_words = (n + 31) // 32
    starts = {i: k for k, (i, _) in enumerate(chains)}
    cin = lambda k: f"S{n_state + k // 32}"
    cout = lambda k: f"S{n_state + n_words + k // 32}"
    out = Body()
    open_chain = None
    for i, op in enumerate(body.ops):
        if i in starts:
            if open_chain is not None:
                out.ADDC(cout(open_chain), cout(open_chain), cout(open_chain))
            k = starts[i]
            out.ADD_CO(cin(k), cin(k), cin(k))
            open_chain = k
            if op.kind == "add_co":
                out.ops.append(Op("addc", op.dst, op.srcs, 0))
            continue
        out.ops.append(op)
    if open_chain is not None:
        out.ADDC(cout(open_chain), cout(open_chain), cout(open_chain))
    if n % 32:
        last = f"S{n_state + 2 * n_words - 1}"
        out.LSHL_IMM(last, last, 32 - n % 32)
    return out, [c for _, c in chains], n_words


def bitpal_packed_block_body(nw: int, sc: BitpalScores = BITPAL_DEFAULT):
    """Packed-carry column-block form of bitpal_body: (body, initial carry-in per chain, carry words per direction)."""
    body, init, n_words = make_blocked_packed(bitpal_body(nw, sc), sc.planes * nw)
    return schedule(body, 200), init, n_words


def bitpal_packed_blocked_simulate(subjects: np.ndarray, query: np.ndarray, nw_block: int,
                                   sc: BitpalScores = BITPAL_DEFAULT, semi: bool = False) -> np.ndarray:
    """bitpal_packed_blocked_kernel at toy scale: column blocks of nw_block words, per query row the carry words of
    the block to the left are this block's carry-in words and its own carry-out words replace them."""
    n, slen = subjects.shape
    qlen = len(query)
    B = sc.planes
    nw_total = (slen + 31) // 32
    n_blocks = (nw_total + nw_block - 1) // nw_block
    peq = build_peq32(subjects, n_blocks * nw_block)
    body, init, n_words = bitpal_packed_block_body(nw_block, sc)
    assert not any(init) and n_words == (sc.chains + 31) // 32
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
    carry = [[np.zeros(n, np.uint32) for _ in range(n_words)] for _ in range(qlen)]   # [row][word]
    run = np.full(n, sc.gap * qlen, dtype=np.int64)
    best = run.copy()
    score = np.full(n, sc.gap * (qlen + slen), dtype=np.int64)
    base = B * nw_block
    for blk in range(n_blocks):
        st = bitpal_init_state(nw_block, n, sc, semi) + [np.zeros(n, np.uint32) for _ in range(2 * n_words)]
        for r, ch in enumerate(query):
            for j in range(n_words):
                st[base + j] = carry[r][j].copy()
            c = code.get(int(ch), 0)
            body.simulate(st, [peq[c, blk * nw_block + w] for w in range(nw_block)])
            carry[r] = [st[base + n_words + j].copy() for j in range(n_words)]
        for w in range(nw_block):
            cols = min(32, max(0, slen - 32 * (blk * nw_block + w)))
            if cols == 0:
                continue
            vals = bitpal_column_values(st, w, sc)[:, :cols]
            steps = run[:, None] + np.cumsum(vals + sc.gap, axis=1)
            best = np.maximum(best, steps.max(axis=1))
            run = steps[:, -1]
            score += vals.sum(axis=1)
    return (best if semi else score).astype(np.int16)


def bitpal_block_body(nw: int, sc: BitpalScores = BITPAL_DEFAULT):
    """Column-block form of bitpal_body: (body, initial carry-in per chain)."""
    body, init = make_blocked(bitpal_body(nw, sc), sc.planes * nw)
    return schedule(body, 200), init


def bitpal_body(nw: int, sc: BitpalScores = BITPAL_DEFAULT) -> Body:
    """For the default scores this computes what the packed kernel of original/BGSA_AVX2/align_core.c:
    183-428 computes (194 ALU operations per word there), on four unsigned planes, in 69 fast-class VALU."""
    return bitpal_scores_body(nw, sc)


def bitpal_init_state(nw: int, lanes: int, sc: BitpalScores = BITPAL_DEFAULT, semi: bool = False) -> list:
    return bitpal_scores_init_state(nw, lanes, sc, semi)


def bitpal_score(state: list, nw: int, qlen: int, slen: int, sc: BitpalScores = BITPAL_DEFAULT,
                 semi: bool = False) -> np.ndarray:
    return bitpal_scores_score(state, nw, qlen, slen, sc, semi)


def bitpal_blocked_simulate(subjects: np.ndarray, query: np.ndarray, nw_block: int,
                            sc: BitpalScores = BITPAL_DEFAULT, semi: bool = False) -> np.ndarray:
    """BitPAl over column blocks of nw_block words with carry words between blocks — the scheme of
    bitpal_blocked_kernel at toy scale.  Returns int16."""
    n, slen = subjects.shape
    qlen = len(query)
    B = sc.planes
    nw_total = (slen + 31) // 32
    n_blocks = (nw_total + nw_block - 1) // nw_block
    peq = build_peq32(subjects, n_blocks * nw_block)
    body, init = bitpal_block_body(nw_block, sc)
    n_ch = len(init)
    assert n_ch == sc.chains and not any(init)
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
    n_chunks = (qlen + 31) // 32
    carry = [[np.zeros(n, np.uint32) for k in range(n_ch)] for _ in range(n_chunks)]
    run = np.full(n, sc.gap * qlen, dtype=np.int64)      # S[m][j] walking right along the last row
    best = run.copy()
    base = B * nw_block
    for blk in range(n_blocks):
        st = bitpal_init_state(nw_block, n, sc, semi) + [c.copy() for c in carry[0]] + \
             [np.zeros(n, np.uint32) for _ in range(n_ch)]
        for r, ch in enumerate(query):
            if r > 0 and r % 32 == 0:
                j = r // 32
                carry[j - 1] = [st[base + n_ch + k].copy() for k in range(n_ch)]
                for k in range(n_ch):
                    st[base + k] = carry[j][k].copy()
            c = code.get(int(ch), 0)
            body.simulate(st, [peq[c, blk * nw_block + w] for w in range(nw_block)])
        tail = qlen % 32
        last = [st[base + n_ch + k] for k in range(n_ch)]
        if tail:
            last = [x << np.uint32(32 - tail) for x in last]
        carry[n_chunks - 1] = [x.copy() for x in last]
        for w in range(nw_block):
            cols = min(32, max(0, slen - 32 * (blk * nw_block + w)))
            if cols == 0:
                continue
            steps = run[:, None] + np.cumsum(bitpal_column_values(st, w, sc)[:, :cols] + sc.gap, axis=1)
            best = np.maximum(best, steps.max(axis=1))
            run = steps[:, -1]
    return (best if semi else run).astype(np.int16)


# =================================================================================================
# Reference-style match masks for the simulator (32 data bits per word)
# =================================================================================================

def build_peq32(subjects: np.ndarray, nw: int) -> np.ndarray:
    """[5][nw][n_subjects] uint32; alphabet map of original/BGSA_CPU/global.c:9-15."""
    n, slen = subjects.shape
    code = np.zeros(256, dtype=np.uint8)
    for ch, c in zip(b"ACGTN", range(5)):
        code[ch] = c
    mapped = code[subjects]
    peq = np.zeros((5, nw, n), dtype=np.uint32)
    for p in range(slen):
        for c in range(5):
            peq[c, p // 32] |= ((mapped[:, p] == c).astype(np.uint32) << np.uint32(p % 32))
    return peq


def run_rows(body: Body, state: list, peq: np.ndarray, query: np.ndarray, groups: int = 1) -> None:
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}
    nw = peq.shape[1]
    planes = code_planes(peq)
    for ch in query:
        c = code.get(int(ch), 0)
        eq = [peq[c, w] for _ in range(groups) for w in range(nw)]
        body.simulate(state, eq, cls=c, planes=planes)
