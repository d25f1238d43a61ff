"""Structured front end for the hand-scheduled gfx950 (CDNA4) kernels of libhvlm_hip.so.

Why this exists: the attention inner loops need one wave per SIMD with the whole 512-entry register file allocated BY HAND
(accumulators, Q, K and V fragments in AccVGPRs, the softmax in the arch VGPRs) and an instruction stream in which every MFMA
is followed by a counted number of VALU / LDS / DMA "fillers".  hipcc's register allocator cannot be told either (DESIGN.md,
"attention, round 4": the C++ pipeline spilled or shuffled ~200 values per tile through AccVGPRs), so these kernels are
emitted as assembly text by Python generators built on this module and assembled with clang (-x assembler -mcpu=gfx950).

What the module does that raw text would not:
  * registers are objects (V / A / S ranges) handed out by named allocators — a kernel cannot alias two live values by typo;
  * every instruction records its definitions and uses, so a HAZARD pass can enforce the software-visible wait states of
    gfx950 that hipcc pads for compiled code and nobody pads for assembly (measured with hipcc -save-temps on probe kernels,
    profiles/r05a_hazard_probe.txt):
        MFMA 32x32x16 (8 pass) result -> any non-accumulate reader / writer ......... 12 wait states
        VALU result (incl. v_accvgpr_write, v_cvt_pk) -> MFMA operand ............... 2
        transcendental / v_cvt_pk result -> VALU reader .............................. 1
        VALU result -> v_permlane32_swap operand ..................................... 2
        s_mov m0 -> LDS-DMA (buffer_load ... lds) .................................... 1
        VALU-written SGPR / VCC -> VMEM reading it ................................... 5
    The pass inserts s_nop where a schedule violates one and counts them (a good schedule needs none in the loop);
  * LDS reads are tracked in issue order, so the counted `s_waitcnt lgkmcnt(N)` in front of the first consumer of a fragment is
    derived, not typed;
  * `interleave` merges an MFMA stream with filler streams gap by gap (the "<= 5 issues per 32x32x16 gap" budget of
    MI355X_MICROARCH.md "Per-instruction cycle constants").
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field


# ---------------------------------------------------------------------------------------------------------------------
# registers
# ---------------------------------------------------------------------------------------------------------------------
class Reg:
    """`n` consecutive registers of one file: kind 'v' (arch VGPR), 'a' (AccVGPR) or 's' (SGPR)."""
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind: str, idx: int, n: int = 1):
        assert kind in ("v", "a", "s") and idx >= 0 and n >= 1
        self.kind, self.idx, self.n = kind, idx, n

    def __str__(self):
        return f"{self.kind}{self.idx}" if self.n == 1 else f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    __repr__ = __str__

    def __getitem__(self, i):
        if isinstance(i, slice):
            a, b, st = i.indices(self.n)
            assert st == 1 and b > a
            return Reg(self.kind, self.idx + a, b - a)
        assert 0 <= i < self.n, (self, i)
        return Reg(self.kind, self.idx + i, 1)

    def sub(self, off: int, n: int) -> "Reg":
        assert 0 <= off and off + n <= self.n, (self, off, n)
        return Reg(self.kind, self.idx + off, n)

    def ids(self):
        return {(self.kind, self.idx + i) for i in range(self.n)}


class Special:
    """vcc / m0 / exec / scc: named, tracked as one pseudo register each."""
    def __init__(self, name):
        self.name = name

    def __str__(self):
        return self.name

    def ids(self):
        return {("x", self.name)}


VCC, M0, EXEC = Special("vcc"), Special("m0"), Special("exec")


class RegFile:
    """Bump allocator with names (for the listing) and an alignment argument (64-bit SGPR pairs, descriptor quads)."""
    def __init__(self, kind: str, start: int = 0, limit: int = 256):
        self.kind, self.next, self.limit, self.names = kind, start, limit, []

    def alloc(self, name: str, n: int = 1, align: int = 1) -> Reg:
        self.next = (self.next + align - 1) // align * align
        r = Reg(self.kind, self.next, n)
        self.next += n
        assert self.next <= self.limit, f"out of {self.kind} registers at {name}"
        self.names.append((name, r))
        return r

    def listing(self):
        return "\n".join(f"//   {str(r):14s} {name}" for name, r in self.names)


# ---------------------------------------------------------------------------------------------------------------------
# instructions
# ---------------------------------------------------------------------------------------------------------------------
TRANS_OPS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32"}


@dataclass
class Ins:
    op: str
    ops: list                      # operands in assembly order (Reg / Special / int / float / str)
    ndef: int = 1                  # how many leading operands are written
    mods: str = ""                 # trailing modifiers ("offset:16", "offen lds", ...)
    xdef: list = field(default_factory=list)   # implicit definitions (vcc, m0, exec)
    xuse: list = field(default_factory=list)   # implicit uses
    rw: bool = False               # the first operand is read AND written (v_fmac, permlane swap, accumulate-in-place helpers)
    comment: str = ""

    # -- classification ------------------------------------------------------------------------------------------
    @property
    def is_mfma(self): return self.op.startswith("v_mfma")
    @property
    def is_valu(self): return self.op.startswith("v_") and not self.is_mfma
    @property
    def is_trans(self): return self.op in TRANS_OPS
    @property
    def is_cvtpk(self): return self.op.startswith("v_cvt_pk")
    @property
    def is_perm(self): return self.op.startswith("v_permlane")
    @property
    def is_lds(self): return self.op.startswith("ds_")
    @property
    def is_vmem(self): return self.op.startswith("buffer_") or self.op.startswith("global_")
    @property
    def is_ldsdma(self): return self.is_vmem and " lds" in (" " + self.mods)
    @property
    def is_nop(self): return self.op == "s_nop"
    @property
    def states(self):              # issue slots this instruction occupies for the wait-state arithmetic
        return (int(self.ops[0]) + 1) if self.is_nop else 1

    def defs(self):
        d = set()
        for o in self.ops[: self.ndef]:
            if hasattr(o, "ids"): d |= o.ids()
        for o in self.xdef: d |= o.ids()
        return d

    def uses(self):
        u = set()
        for o in self.ops[self.ndef:]:
            if hasattr(o, "ids"): u |= o.ids()
        if self.rw and self.ops and hasattr(self.ops[0], "ids"): u |= self.ops[0].ids()
        for o in self.xuse: u |= o.ids()
        if self.is_perm:                        # a swap reads and writes both operands
            for o in self.ops:
                if hasattr(o, "ids"): u |= o.ids()
        return u

    def all_defs(self):
        d = self.defs()
        if self.is_perm:
            for o in self.ops:
                if hasattr(o, "ids"): d |= o.ids()
        return d

    def text(self):
        def fmt(o):
            if isinstance(o, float): return repr(o)
            if isinstance(o, int) and not isinstance(o, bool) and (o > 64 or o < -16): return hex(o & 0xffffffff)
            return str(o)
        s = self.op
        if self.ops: s += " " + ", ".join(fmt(o) for o in self.ops)
        if self.mods: s += " " + self.mods
        return s


@dataclass
class Label:
    name: str


@dataclass
class Raw:
    text: str


class Program:
    """A list of Ins / Label / Raw with emit helpers.  `ins()` appends; `seq()` builds a detached list for the scheduler."""
    def __init__(self):
        self.items: list = []

    # generic
    def add(self, x):
        if isinstance(x, (list, tuple)):
            for y in x: self.add(y)
        else:
            self.items.append(x)
        return x

    def label(self, name): self.items.append(Label(name))
    def raw(self, text): self.items.append(Raw(text))
    def comment(self, text): self.items.append(Raw("// " + text))


def I(op, *ops, ndef=1, mods="", xdef=(), xuse=(), rw=False, comment=""):
    return Ins(op, list(ops), ndef, mods, list(xdef), list(xuse), rw, comment)


# convenience constructors ------------------------------------------------------------------------------------------------
def s_nop(n): return I("s_nop", n, ndef=0)
def s_waitcnt(vm=None, lgkm=None):
    parts = []
    if vm is not None: parts.append(f"vmcnt({vm})")
    if lgkm is not None: parts.append(f"lgkmcnt({lgkm})")
    return I("s_waitcnt", " ".join(parts), ndef=0)
def mfma_bf16(d, a, b, c):
    """v_mfma_f32_32x32x16_bf16 d[16], a[4], b[4], c (Reg[16] or 0)."""
    assert d.n == 16 and a.n == 4 and b.n == 4
    return I("v_mfma_f32_32x32x16_bf16", d, a, b, c)


# ---------------------------------------------------------------------------------------------------------------------
# hazard pass
# ---------------------------------------------------------------------------------------------------------------------
MFMA_RESULT_STATES = 12        # 8-pass XDL write -> VALU / LDS / VMEM / non-accumulate MFMA use (hipcc: s_nop 11)
VALU_TO_MFMA = 2
TRANS_TO_VALU = 1
VALU_TO_PERM = 2
M0_TO_LDSDMA = 1
VSGPR_TO_VMEM = 5
DOT_TO_VALU_READ = 3         # v_dot2 result -> a different VALU reading it (or the same opcode reading it as A / B)
DOT_TO_VALU_WRITE = 4        # v_dot2 result -> a different VALU overwriting it
WINDOW = 16


def _need(prev: Ins, cur: Ins) -> int:
    """wait states required between `prev` and a LATER `cur` (0 = none)."""
    need = 0
    pd = prev.all_defs()
    if not pd: return 0
    cu, cd = cur.uses(), cur.all_defs()
    if prev.is_mfma:
        ov = pd & (cu | cd)
        if ov:
            acc_chain = (cur.is_mfma and len(cur.ops) == 4 and hasattr(cur.ops[3], "ids") and cur.ops[3].ids() == prev.ops[0].ids()
                         and cur.ops[0].ids() == prev.ops[0].ids()
                         and not (pd & (cur.ops[1].ids() | cur.ops[2].ids())))
            # an MFMA that overwrites the block without reading it (C = 0) still has to respect the in-order pipe: no software
            # wait is listed for XDL write -> XDL write of the same vDst, the pipe retires in order
            overwrite = (cur.is_mfma and cur.ops[0].ids() == prev.ops[0].ids() and not (pd & cu))
            if not acc_chain and not overwrite:
                need = max(need, MFMA_RESULT_STATES)
    if prev.op.startswith("v_dot"):
        # DOT results (LLVM GCNHazardRecognizer, gfx90a+ rules: DotWriteSameDotReadSrcAB = 3, DotWriteDifferentVALURead = 3,
        # DotWriteDifferentVALUWrite = 4): back-to-back accumulation through SrcC of the SAME opcode is free, everything else waits
        same_acc = (cur.op == prev.op and len(cur.ops) == 4 and hasattr(cur.ops[3], "ids") and cur.ops[3].ids() == prev.ops[0].ids()
                    and not (pd & (cur.ops[1].ids() if hasattr(cur.ops[1], "ids") else set()))
                    and not (pd & (cur.ops[2].ids() if hasattr(cur.ops[2], "ids") else set())))
        if (cur.is_valu or cur.is_mfma) and (pd & cu) and not same_acc: need = max(need, DOT_TO_VALU_READ)
        if (cur.is_valu or cur.is_mfma) and (pd & cd) and not same_acc: need = max(need, DOT_TO_VALU_WRITE)
    if prev.is_valu:
        if cur.is_mfma and (pd & cu): need = max(need, VALU_TO_MFMA)
        if (prev.is_trans or prev.is_cvtpk) and cur.is_valu and (pd & cu): need = max(need, TRANS_TO_VALU)
        if cur.is_perm and (pd & (cu | cd)): need = max(need, VALU_TO_PERM)
        sg = {x for x in pd if x[0] in ("s", "x")}
        if sg and cur.is_vmem and (sg & cu): need = max(need, VSGPR_TO_VMEM)
    if prev.op in ("s_mov_b32", "s_add_u32", "s_add_i32") and ("x", "m0") in pd and cur.is_ldsdma:
        need = max(need, M0_TO_LDSDMA)
    return need


def fix_hazards(items: list, tag: str = "", report: list | None = None, carry: list | None = None) -> list:
    """Insert s_nop so that every rule of `_need` holds on the straight-line order of `items` (labels do not reset the
    window: fall-through is the common path and the conservative choice).  `carry`: instructions that precede items[0]
    on some path (e.g. the tail of a loop body for its own head) — they are checked, not emitted."""
    out: list = []
    hist: list[tuple[Ins, int]] = []      # (instruction, states issued AFTER it so far)
    for p in (carry or []):
        if isinstance(p, Ins):
            hist = [(q, s + p.states) for q, s in hist][-WINDOW:]
            hist.append((p, 0))
    nops = 0
    for it in items:
        if not isinstance(it, Ins):
            out.append(it)
            continue
        need = 0
        for prev, gone in hist:
            n = _need(prev, it)
            if n > gone: need = max(need, n - gone)
        if need:
            out.append(s_nop(need - 1))
            nops += need
            hist = [(q, s + need) for q, s in hist]
        out.append(it)
        hist = [(q, s + it.states) for q, s in hist if s + it.states <= WINDOW + 4][-WINDOW * 2:]
        hist.append((it, 0))
    if report is not None: report.append((tag, nops))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# LDS read tracker: derive the counted lgkmcnt waits
# ---------------------------------------------------------------------------------------------------------------------
class LgkmTracker:
    """LDS operations of one wave return IN ORDER, so `s_waitcnt lgkmcnt(N)` = "all but the N youngest have returned".
    The counter is 4 bits wide; with more than 15 younger operations behind a read, lgkmcnt(15) already implies it."""
    MAXCNT = 15

    def __init__(self, initial: list | None = None):
        self.pending: list[set] = list(initial or [])     # definitions of the outstanding reads, oldest first

    def feed(self, it: Ins, out: list):
        touched = it.uses() | it.all_defs()
        # oldest .. youngest: find the youngest pending read this instruction depends on
        hit = -1
        for i, d in enumerate(self.pending):
            if d & touched: hit = i
        if hit >= 0:
            younger = len(self.pending) - 1 - hit
            out.append(s_waitcnt(lgkm=min(younger, self.MAXCNT)))
            self.pending = self.pending[hit + 1:] if younger <= self.MAXCNT else self.pending[-self.MAXCNT:]
        if it.op == "s_waitcnt" and "lgkmcnt" in str(it.ops[0]):
            n = int(re.search(r"lgkmcnt\((\d+)\)", str(it.ops[0])).group(1))
            self.pending = self.pending[len(self.pending) - n:] if n < len(self.pending) else self.pending
        if it.is_lds:
            self.pending.append(it.defs() if it.op.startswith("ds_read") else set())
        out.append(it)


def add_lgkm_waits(items: list, initial: list | None = None) -> tuple[list, list]:
    tr = LgkmTracker(initial)
    out: list = []
    for it in items:
        if isinstance(it, Ins): tr.feed(it, out)
        else: out.append(it)
    return out, tr.pending


# ---------------------------------------------------------------------------------------------------------------------
# MFMA / filler interleave
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class Filler:
    """A group of instructions that must stay together and in order, placed no earlier than the gap after MFMA `after`
    (-1 = before the first MFMA) and, if `before` is set, no later than the gap after MFMA `before - 1`."""
    ins: list
    after: int = -1
    before: int | None = None
    prio: int = 1              # 0 = place as early as allowed (LDS reloads, DMA), 1 = spread evenly (VALU)
    cost: float | None = None  # issue slots; default = number of instructions (transcendentals count 5/3)

    def weight(self):
        if self.cost is not None: return self.cost
        w = 0.0
        for i in self.ins:
            if isinstance(i, Ins): w += (5.0 / 3.0) if i.is_trans else (0.0 if i.is_nop else 1.0)
        return w


def interleave(mfmas: list, early: list, stream: list, head_room: int = 0) -> list:
    """One step body: `mfmas` in order; `early` fillers (prio 0) go into the first gap they are allowed in, at most a few per
    gap; `stream` (prio 1, IN ORDER) is spread so that every gap carries about the same issue weight.  `head_room`: number
    of leading gaps that take no `stream` filler (results of the previous step's last MFMAs are not readable yet)."""
    n = len(mfmas)
    gaps: list[list] = [[] for _ in range(n + 1)]          # gaps[0] = before MFMA 0, gaps[g + 1] = after MFMA g
    load = [0.0] * (n + 1)
    EARLY_CAP = 3.0
    for f in sorted(early, key=lambda f: f.after):
        g = f.after + 1
        while g < n and load[g] + f.weight() > EARLY_CAP and (f.before is None or g + 1 < f.before + 1): g += 1
        g = min(g, n)
        gaps[g].append(f)
        load[g] += f.weight()
    total = sum(f.weight() for f in stream) + sum(load)
    usable = max(1, n - head_room)
    target = total / usable
    g = 1 + head_room if n else 0
    for f in stream:
        g = max(g, f.after + 1)
        while g < n and load[g] + 0.5 * f.weight() > target: g += 1
        g = min(g, n)
        gaps[g].append(f)
        load[g] += f.weight()
    out: list = []
    for x in gaps[0]: out += x.ins
    for i, m in enumerate(mfmas):
        out.append(m)
        for x in gaps[i + 1]: out += x.ins
    return out


# ---------------------------------------------------------------------------------------------------------------------
# text output
# ---------------------------------------------------------------------------------------------------------------------
def render(items: list) -> str:
    lines = []
    for it in items:
        if isinstance(it, Ins):
            lines.append("  " + it.text() + (f"    // {it.comment}" if it.comment else ""))
        elif isinstance(it, Label):
            lines.append(f"{it.name}:")
        else:
            lines.append("  " + it.text)
    return "\n".join(lines) + "\n"


def kernel_text(name: str, body: str, *, lds_bytes: int, kernarg_bytes: int, n_arch_vgpr: int, n_acc_vgpr: int, n_sgpr: int,
                args_meta: list[tuple[int, int, str]], header_comment: str = "") -> str:
    """Wrap `body` into a complete code-object source: kernel descriptor + metadata (code object v5 conventions).
    args_meta: (size, offset, kind) with kind in {"global_buffer", "by_value"}."""
    accum_offset = (n_arch_vgpr + 3) // 4 * 4
    total_vgpr = accum_offset + n_acc_vgpr
    assert total_vgpr <= 512
    args = "\n".join(
        f"      - {{.size: {sz}, .offset: {off}, .value_kind: {kind}" + (", .address_space: global" if kind == "global_buffer" else "") + "}"
        for sz, off, kind in args_meta)
    return f"""{header_comment}
.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
.amdhsa_code_object_version 5
.text
.globl {name}
.p2align 8
.type {name},@function
{name}:
{body}
.Lend_{name}:
.size {name}, .Lend_{name} - {name}

.rodata
.p2align 6
.amdhsa_kernel {name}
  .amdhsa_group_segment_fixed_size {lds_bytes}
  .amdhsa_private_segment_fixed_size 0
  .amdhsa_kernarg_size {kernarg_bytes}
  .amdhsa_user_sgpr_count 2
  .amdhsa_user_sgpr_kernarg_segment_ptr 1
  .amdhsa_system_sgpr_workgroup_id_x 1
  .amdhsa_system_sgpr_workgroup_id_y 0
  .amdhsa_system_sgpr_workgroup_id_z 0
  .amdhsa_system_vgpr_workitem_id 0
  .amdhsa_next_free_vgpr {total_vgpr}
  .amdhsa_next_free_sgpr {n_sgpr}
  .amdhsa_accum_offset {accum_offset}
  .amdhsa_reserve_vcc 1
  .amdhsa_float_round_mode_32 0
  .amdhsa_float_round_mode_16_64 0
  .amdhsa_float_denorm_mode_32 3
  .amdhsa_float_denorm_mode_16_64 3
  .amdhsa_dx10_clamp 1
  .amdhsa_ieee_mode 0
  .amdhsa_tg_split 0
.end_amdhsa_kernel

.amdgpu_metadata
---
amdhsa.version: [1, 2]
amdhsa.kernels:
  - .name: {name}
    .symbol: {name}.kd
    .kernarg_segment_size: {kernarg_bytes}
    .group_segment_fixed_size: {lds_bytes}
    .private_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .wavefront_size: 64
    .sgpr_count: {n_sgpr + 6}
    .vgpr_count: {total_vgpr}
    .agpr_count: {n_acc_vgpr}
    .max_flat_workgroup_size: 256
    .args:
{args}
...
.end_amdgpu_metadata
"""
