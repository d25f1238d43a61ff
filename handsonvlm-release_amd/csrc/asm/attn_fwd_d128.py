#!/usr/bin/env python3
"""Causal flash-attention FORWARD for dh = 128, bf16, gfx950 — hand-allocated registers, one wave per SIMD, persistent.

Replaces, for the decoder's shapes, attn_fwd_kernel<128, true, 4> of attention.hip (same operands, same results to rounding):
    softmax(Q K^T / sqrt(dh) + causal mask) V          (transformers LlamaAttention; the flash path of the reference,
                                                        llava/train/llama_flash_attn_monkey_patch.py:81-94, incl. cu_seqlens)
reading q / k / v straight from the token-major fused buffer qkv[tokens, ld] and writing O[tokens, H*128] and lse[B, H, S_max].

Structure (why it is not the C++ kernel re-typed):
  * workgroup = 4 waves = ONE wave per SIMD, 512 registers each; a work ITEM is 256 query rows of one (sample, head) = eight
    32-row blocks, and wave w owns blocks w and w + 4 ("qb0", "qb1"): near the diagonal every wave still has work, so the
    lock-step K/V stream wastes 3 of (16 j + 12) half-units instead of a whole wave;
  * per wave:  O (2 x 64) , Q (2 x 32), K fragments (32) and V fragments (32) live in AccVGPRs — exactly 256 — and the scores,
    probabilities, softmax state and the NEXT item's Q in ~230 arch VGPRs;
  * a step = 32 keys.  Three-stage software pipeline with NO dependency inside a step:
        MFMA stream :  QK^T of step s+2  (-> S[p])   alternating with   P V of step s  (<- P[p])
        VALU stream :  softmax of step s+1  (S[1-p] -> P[1-p], running max / sum)
        LDS stream  :  K fragments of step s+3 and V fragments of step s+1, each reloaded right after its last use
    so the matrix pipe never waits for the exponentials of its own tile (what bounded the C++ kernels: r03d_valu_mfma_probe);
  * the running maximum is only RAISED when some score exceeds it by 2^THR (THR = 8): the O rescale (128 AccVGPRs through the
    VALU) is an out-of-line slow path, applied at the next step boundary when every P V of the old scale has been issued;
  * K / V tiles (64 keys) arrive by LDS-DMA into a THREE-slot ring, stage X = {K tile X+1, V tile X}, issued two iterations
    ahead of their first read (LDS-DMA takes ~3400 cycles issue -> landed under load, profiles/r05b): one barrier per tile and
    a counted vmcnt.  K's LDS image is [half][ks][hi][key] so that a fragment read is base + immediate; V keeps the
    XOR-swizzled row-major image that ds_read_b64_tr_b16 needs;
  * PERSISTENT: one workgroup per CU pulls items from a per-XCD queue (heaviest block of a group of 8 (sample, head) pairs first:
    the pairs of a group share one L2) and, during the last iterations of an item, issues the NEXT item's first K tile, its
    first three stages and its Q rows — version 1 spent 44 % of its cycles in per-workgroup prologue loads and epilogues.

The file is a GENERATOR: `python3 attn_fwd_d128.py out.s` writes the assembly (asmgen.py pads hazards and derives the lgkmcnt
waits); the Makefile assembles it into a code object that attention_asm.hip embeds and launches through hipModuleLaunchKernel.
"""
import re
import sys
from asmgen import (Reg, RegFile, Program, I, Ins, Label, Raw, Filler, VCC, M0, EXEC, s_nop, s_waitcnt, mfma_bf16, interleave,
                    fix_hazards, add_lgkm_waits, render, kernel_text)

KNAME = "hvlm_fa_fwd_d128_causal"
THR_BITS = 0x41000000          # 8.0f: the running maximum is raised only when a score exceeds it by more than 2^8
NEG_INF = 0xff800000
LN2 = 0x3f317218

# LDS map -------------------------------------------------------------------------------------------------------------
SLOT = 32768                   # stage = K tile (16 KiB) + V tile (16 KiB); three slots
V_IN_SLOT = 16384
K0AREA = 3 * SLOT              # the first K tile of an item (read by its prologue only)
OSTAGE = K0AREA + 16384        # 4 x 8 KiB wave-private blocks for the O epilogue
BCAST = OSTAGE + 4 * 8192      # two words: the queue index of the next item, wave 0 -> everybody
LDS_BYTES = BCAST + 64

# kernel arguments ------------------------------------------------------------------------------------------------------
ARGS = [(8, 0, "global_buffer"), (8, 8, "global_buffer"), (8, 16, "global_buffer"), (8, 24, "global_buffer"),
        (4, 32, "by_value"), (4, 36, "by_value"), (4, 40, "by_value"), (4, 44, "by_value"),
        (4, 48, "by_value"), (4, 52, "by_value"), (4, 56, "by_value"), (4, 60, "by_value"),
        (8, 64, "global_buffer"), (4, 72, "by_value"), (4, 76, "by_value")]
KERNARG_BYTES = 80


class Gen:
    def __init__(self, debug_serial=False, abl=(), prof=False, k_rows=True, dot_sum=False, lazy_max=False):
        """abl: timing-only ablations (WRONG RESULTS): 'nosm' no softmax stream, 'nodma' no LDS-DMA in the loop, 'nolds' no
        fragment reloads, 'nomfma' no MFMAs.  prof: s_memtime cycle sums per workgroup phase through the extra kernel argument."""
        self.debug_serial = debug_serial
        self.abl, self.prof = set(abl), prof
        # k_rows: K tiles keep the row-major XOR-swizzled image of V (DMA lanes fetch whole 256-byte rows) instead of the
        # [half][ks][hi][key] image whose fragment address is base + immediate but whose DMA fetches 32-byte granules of 32 rows per
        # instruction (the CLIP kernel, which is HBM-bound, measured -14 % with whole rows: profiles/r05f).  Measured HERE on a settled chip
        # (1.5 s loops, time and joules per call, profiles/r05z_attn_variant_energy.txt): row image 163.0-164.6 us / 226-228 mJ against 166.9-167.5 us /
        # 234-235 mJ for the granule image at S = 867, 201 vs 206 us at S = 2048: the row image is the default (dev variant `kgran` = the granule
        # image).  A first A/B in 20-launch timing loops (profiles/r05h) had ranked them the other way round: such loops do not settle the chip.
        self.k_rows = k_rows
        self.lazy_max = lazy_max    # slow-path trigger on the lane's PARTIAL step maximum in raw-score units: no half-wave exchange, no scaling on the fast path
        self.dot_sum = dot_sum      # row sum as v_dot2_f32_bf16 over the ROUNDED probability pairs (8 per block instead of 16 v_add_f32 on the fp32 values)
        self.report = []
        self.site = 0
        self.ool = []                     # out-of-line blocks (slow paths), appended behind the main code
        S = self.S = RegFile("s", 0, 102)
        V = self.V = RegFile("v", 0, 256)
        A = self.A = RegFile("a", 0, 256)
        # ---- SGPRs ------------------------------------------------------------------------------------------------
        self.s_karg = S.alloc("kernarg", 2)
        self.s_wg = S.alloc("workgroup id", 1)
        self.s_bc = S.alloc("broadcast word in use (0 / 4)", 1)
        self.s_rq = S.alloc("rsrc qkv (this item's sample rows)", 4, 4)
        self.s_args0 = S.alloc("args 0..7: qkv, out, lse, cu", 8, 4)
        self.s_args1 = S.alloc("args 8..15: S_max H ld scale nqb BH magic_nqb magic_H", 8, 4)
        self.s_args2 = S.alloc("args: queue, G, pad", 4, 4)
        self.s_ro = S.alloc("rsrc out (sample rows)", 4, 4)
        self.s_rd = S.alloc("rsrc of the DMA stream (this item's, then the next item's)", 4, 4)
        self.s_rn = S.alloc("rsrc qkv of the NEXT item", 4, 4)
        self.s_qkv, self.s_out, self.s_lse, self.s_cu = (self.s_args0.sub(0, 2), self.s_args0.sub(2, 2), self.s_args0.sub(4, 2),
                                                         self.s_args0.sub(6, 2))
        (self.s_smax, self.s_H, self.s_ld, self.s_scale, self.s_nqb, self.s_BH, self.s_magq, self.s_magh) = [self.s_args1[i] for i in range(8)]
        self.s_queue = self.s_args2.sub(0, 2); self.s_G = self.s_args2[2]; self.s_cs0 = self.s_args2[3]   # (pad word reused: ring slot at item start)
        self.s_t = [S.alloc(f"tmp{i}", 1) for i in range(8)]
        self.s_t64 = S.alloc("tmp64 / mask64", 2, 2)
        self.s_exec = S.alloc("saved exec", 2, 2)
        self.s_m64 = self.s_t64                      # (the s_load pair of emit_params and the epilogue's lane mask are never live together)
        self.s_h = S.alloc("h", 1); self.s_j = S.alloc("j (256-row block)", 1)
        self.s_S = S.alloc("S (this sample)", 1)
        self.s_w = S.alloc("wave", 1)
        self.s_n0 = S.alloc("n0: steps of qb0", 1); self.s_n1 = S.alloc("n1: steps of qb1", 1)
        self.s_rb0 = S.alloc("first row of qb0", 1); self.s_rb1 = S.alloc("first row of qb1", 1)
        self.s_nT = S.alloc("key tiles of the item", 1)
        self.s_U = S.alloc("iteration", 1); self.s_step = S.alloc("step", 1)
        self.s_pend = S.alloc("pending O rescale bits", 1)
        self.s_ld2 = S.alloc("ld * 2 bytes", 1)
        self.s_kcol = S.alloc("K soffset base of this wave", 1); self.s_vcol = S.alloc("V soffset base", 1)
        self.s_tile = S.alloc("64 * ld * 2", 1); self.s_r16 = S.alloc("16 * ld * 2", 1)
        self.s_ktile = S.alloc("DMA stream: soffset of the next K tile", 1); self.s_vtile = S.alloc("DMA stream: next V tile", 1)
        self.s_m0k = S.alloc("DMA stream: m0 of this wave's first K piece", 1); self.s_m0v = S.alloc("DMA stream: first V piece", 1)
        self.s_dslot = S.alloc("DMA stream: ring slot", 1)
        self.s_dso = S.alloc("DMA stream: next own stage", 1); self.s_nu = S.alloc("DMA stream: next-item units issued", 1)
        self.s_numrec = S.alloc("num_records qkv", 1)
        self.s_so = S.alloc("soffset tmp", 1)
        self.s_lsebase = S.alloc("(b*H+h)*S_max", 1)
        self.s_outrow = S.alloc("H*256", 1)
        self.s_cs = S.alloc("consumer ring slot", 1)
        self.s_li = S.alloc("queue index of this item", 1); self.s_lin = S.alloc("queue index of the next item", 1)
        self.s_items = S.alloc("items of this XCD's queue", 1); self.s_nwg = S.alloc("workgroups on this queue", 1)
        self.s_bh0 = S.alloc("first (sample, head) pair of this XCD", 1)
        self.s_split = S.alloc("queue index where the last (short) group starts", 1); self.s_rem = S.alloc("pairs in the last group", 1)
        self.s_qx = S.alloc("this XCD's queue counter", 2, 2)
        self.s_kcoln = S.alloc("next item: K soffset base", 1); self.s_vcoln = S.alloc("next item: V soffset base", 1)
        self.s_rb0n = S.alloc("next item: first row of qb0", 1); self.s_rb1n = S.alloc("next item: first row of qb1", 1)
        self.s_hqn = S.alloc("next item: h * 256", 1)
        self.s_delta = S.alloc("ring address step", 1)
        if prof:
            self.s_prof = S.alloc("prof buffer", 2, 2)
            self.s_tm = self.s_karg                  # (the kernarg pointer is dead once the arguments are loaded)
            self.s_t0 = S.alloc("previous stamp", 1)
        # ---- AccVGPRs: exactly 256 ---------------------------------------------------------------------------------
        self.aO = [A.alloc(f"O qb{q}", 64) for q in range(2)]
        self.aQ = [A.alloc(f"Q qb{q}", 32) for q in range(2)]
        self.aK = A.alloc("K fragments (8 ks x 4)", 32)
        self.aV = A.alloc("V fragments (2 sp x 4 db x 4)", 32)
        # ---- arch VGPRs --------------------------------------------------------------------------------------------
        self.v_tid = V.alloc("tid (v0)", 1)
        self.vS = [[V.alloc(f"S qb{q} p{p}", 16, 4) for p in range(2)] for q in range(2)]
        self.vP = [[V.alloc(f"P qb{q} p{p}", 8, 4) for p in range(2)] for q in range(2)]
        self.vM = [V.alloc(f"m qb{q}", 1) for q in range(2)]
        self.vNEGM = [V.alloc(f"-m qb{q}", 1) for q in range(2)]
        self.vMT = [V.alloc(f"m+THR qb{q}", 1) for q in range(2)]
        if lazy_max:
            self.vMTR = [V.alloc(f"(m+THR-0.5)/scale qb{q}: the trigger threshold in raw-score units", 1) for q in range(2)]
            self.s_rsc = V.alloc("1 / scale_log2 (a VGPR: the SGPR file is full)", 1)
        self.vL = [V.alloc(f"l qb{q}", 1) for q in range(2)]
        self.vALPHA = [V.alloc(f"pending alpha qb{q}", 1) for q in range(2)]
        self.vT = [V.alloc(f"softmax temps qb{q}", 6) for q in range(2)]
        self.vACC = [V.alloc(f"partial sums qb{q}", 4) for q in range(2)]
        self.vNINF = V.alloc("-inf", 1)
        if dot_sum: self.s_ones = V.alloc("bf16 (1.0, 1.0): operand of the row-sum dot products (a VGPR: the SGPR file is full)", 1)
        self.vLIM = V.alloc("l31 - 4 hi (triangular limit)", 1)
        self.vLIMC = V.alloc("limit in use", 1)
        self.vKADDR = V.alloc("K fragment LDS address (lane*16 + slot)", 1)
        self.vKADDRS = [V.alloc(f"K fragment LDS address, row image, ks{ks}", 1) for ks in range(8)] if k_rows else []
        self.vVADDR = [[V.alloc(f"V tr-read address db{db} hf{hf}", 1) for hf in range(2)] for db in range(4)]
        self.vKOFF = V.alloc("K DMA voffset", 1)
        self.vVOFF = V.alloc("V DMA voffset", 1)
        self.vLANE = V.alloc("lane", 1); self.vL31 = V.alloc("lane & 31", 1); self.vHI = V.alloc("lane >> 5", 1)
        self.vATOM = V.alloc("queue atomic result", 1)
        self.vX = V.alloc("scratch", 12, 4)
        self.vQN = V.alloc("Q of the NEXT item (2 qb x 8 ks x 4)", 64, 4)
        self.vPACC = V.alloc("prof: cycle sums (lane k = phase k)", 1)

    # ---------------------------------------------------------------------------------------------------------------
    def new_site(self):
        self.site += 1
        return self.site

    # ---- DMA of one stage (K tile X+1, V tile X) by this wave: 8 pieces, driven by the DMA-stream registers ---------------
    def dma_stage_ins(self):
        g = []
        for i in range(4):
            if self.k_rows:          # piece 4 i + w: rows 16 i + 4 w + (lane >> 4), like the V pieces (s_m0k = slot + w * 1024, s_ktile without the wave part)
                g.append([I("s_add_u32", M0, self.s_m0k, 0x1000 * i) if i else I("s_mov_b32", M0, self.s_m0k),
                          I("s_add_u32", self.s_t[3], self.s_t[3], self.s_r16) if i else I("s_mov_b32", self.s_t[3], self.s_ktile),
                          I("buffer_load_dwordx4", self.vVOFF, self.s_rd, self.s_t[3], ndef=0, mods="offen lds", xuse=[M0])])
                continue
            g.append([I("s_add_u32", M0, self.s_m0k, 0x400 * i) if i else I("s_mov_b32", M0, self.s_m0k),
                      I("s_add_u32", self.s_so, self.s_ktile, 32 * i) if i else I("s_mov_b32", self.s_so, self.s_ktile),
                      I("buffer_load_dwordx4", self.vKOFF, self.s_rd, self.s_so, ndef=0, mods="offen lds", xuse=[M0])])
        for i in range(4):
            g.append([I("s_add_u32", M0, self.s_m0v, 0x1000 * i) if i else I("s_mov_b32", M0, self.s_m0v),
                      I("s_add_u32", self.s_so, self.s_so, self.s_r16) if i else I("s_mov_b32", self.s_so, self.s_vtile),
                      I("buffer_load_dwordx4", self.vVOFF, self.s_rd, self.s_so, ndef=0, mods="offen lds", xuse=[M0])])
        return g

    def dma_k0_ins(self, kcol, rsrc):
        """The item's first K tile -> K0AREA (4 pieces of this wave)."""
        if self.k_rows:
            out = [I("s_lshl_b32", self.s_t[6], self.s_w, 10), I("s_add_u32", self.s_t[6], self.s_t[6], K0AREA), I("s_mov_b32", self.s_so, kcol)]
            for i in range(4):
                out += [I("s_add_u32", M0, self.s_t[6], 0x1000 * i)] + ([I("s_add_u32", self.s_so, self.s_so, self.s_r16)] if i else [s_nop(0)]) + \
                       [I("buffer_load_dwordx4", self.vVOFF, rsrc, self.s_so, ndef=0, mods="offen lds", xuse=[M0])]
            return out
        out = [I("s_lshl_b32", self.s_t[6], self.s_w, 12), I("s_add_u32", self.s_t[6], self.s_t[6], K0AREA)]
        for i in range(4):
            out += [I("s_add_u32", M0, self.s_t[6], 0x400 * i), I("s_add_u32", self.s_so, kcol, 32 * i),
                    I("buffer_load_dwordx4", self.vKOFF, rsrc, self.s_so, ndef=0, mods="offen lds", xuse=[M0])]
        return out

    def dma_advance(self):
        """DMA stream: one stage issued -> next tile, next ring slot."""
        t = self.s_t
        return [I("s_add_u32", self.s_ktile, self.s_ktile, self.s_tile), I("s_add_u32", self.s_vtile, self.s_vtile, self.s_tile),
                I("s_add_u32", self.s_m0k, self.s_m0k, SLOT), I("s_add_u32", self.s_m0v, self.s_m0v, SLOT),
                I("s_add_u32", self.s_dslot, self.s_dslot, 1),
                I("s_cmp_lt_u32", self.s_dslot, 3, ndef=0), I("s_cbranch_scc1", f"L_dadv_{self.new_site()}", ndef=0),
                I("s_mov_b32", self.s_dslot, 0), I("s_sub_u32", self.s_m0k, self.s_m0k, 3 * SLOT), I("s_sub_u32", self.s_m0v, self.s_m0v, 3 * SLOT),
                Label(f"L_dadv_{self.site}")]

    def dma_set_slot(self, slot):
        """m0 bases of the DMA stream from a ring slot index held in `slot`."""
        t = self.s_t
        return [I("s_mov_b32", self.s_dslot, slot), I("s_lshl_b32", t[6], slot, 15),
                I("s_lshl_b32", t[7], self.s_w, 10 if self.k_rows else 12), I("s_add_u32", self.s_m0k, t[6], t[7]),
                I("s_lshl_b32", t[7], self.s_w, 10), I("s_add_u32", t[7], t[7], V_IN_SLOT), I("s_add_u32", self.s_m0v, t[6], t[7])]

    def consumer_advance(self):
        st = self.new_site()
        out = [I("s_mov_b32", self.s_delta, SLOT), I("s_add_u32", self.s_cs, self.s_cs, 1),
               I("s_cmp_lt_u32", self.s_cs, 3, ndef=0), I("s_cbranch_scc1", f"L_cadv_{st}", ndef=0),
               I("s_mov_b32", self.s_cs, 0), I("s_mov_b32", self.s_delta, (-2 * SLOT) & 0xffffffff),
               Label(f"L_cadv_{st}"),
               I("v_add_u32", self.vKADDR, self.s_delta, self.vKADDR)] + [I("v_add_u32", r_, self.s_delta, r_) for r_ in self.vKADDRS]
        for db in range(4):
            for hf in range(2): out.append(I("v_add_u32", self.vVADDR[db][hf], self.s_delta, self.vVADDR[db][hf]))
        return out

    # ---- fragment reads -------------------------------------------------------------------------------------------
    def k_read(self, ks, half, addr=None):
        if self.k_rows:                  # row (32 half + l31) * 256 + (((2 ks + hi) ^ swz(l31)) << 4); addr: scalar added by the caller's register set
            base = self.vKADDRS[ks] if addr is None else addr[ks]
            return I("ds_read_b128", self.aK.sub(4 * ks, 4), base, mods=f"offset:{half * 8192}")
        return I("ds_read_b128", self.aK.sub(4 * ks, 4), addr or self.vKADDR, mods=f"offset:{half * 8192 + ks * 1024}")

    def v_reads(self, sp, db, half):
        off = V_IN_SLOT + half * 8192 + sp * 4096
        base = self.aV.sub(4 * (4 * sp + db), 4)
        return [I("ds_read_b64_tr_b16", base.sub(2 * hf, 2), self.vVADDR[db][hf], mods=f"offset:{off}") for hf in range(2)]

    # ---- MFMAs -------------------------------------------------------------------------------------------------------
    def qk(self, qb, p, ks):
        d = self.vS[qb][p]
        return mfma_bf16(d, self.aK.sub(4 * ks, 4), self.aQ[qb].sub(4 * ks, 4), 0 if ks == 0 else d)

    def pv(self, qb, p, sp, db):
        o = self.aO[qb].sub(16 * db, 16)
        return mfma_bf16(o, self.aV.sub(4 * (4 * sp + db), 4), self.vP[qb][p].sub(4 * sp, 4), o)

    # ---- softmax of one 32-row block on 32 keys: S[qb][p] -> P[qb][p] -----------------------------------------------------
    def sm_stream(self, qb, p, init=False):
        x = self.vS[qb][p]; P = self.vP[qb][p]; t = self.vT[qb]; acc = self.vACC[qb]
        m, negm, mt, l = self.vM[qb], self.vNEGM[qb], self.vMT[qb], self.vL[qb]
        sc = self.s_scale
        F = []
        def f(*ins, **kw): F.append(Filler(list(ins), **kw))
        f(I("v_max3_f32", t[0], x[0], x[1], x[2])); f(I("v_max3_f32", t[1], x[3], x[4], x[5]))
        f(I("v_max3_f32", t[2], x[6], x[7], x[8])); f(I("v_max3_f32", t[3], x[9], x[10], x[11]))
        f(I("v_max3_f32", t[4], x[12], x[13], x[14]))
        f(I("v_max3_f32", t[0], t[0], t[1], t[2])); f(I("v_max3_f32", t[3], t[3], t[4], x[15]))
        f(I("v_max_f32", t[0], t[0], t[3]))
        lazy = self.lazy_max and not init
        full = [I("v_mov_b32", t[1], t[0]),
                # (the hazard pass pads VALU -> permlane; with two qb chains interleaved the other chain's instructions sit in between)
                I("v_permlane32_swap_b32", t[1], t[0], ndef=2),
                I("v_max_f32", t[0], t[0], t[1]),
                I("v_mul_f32", t[0], sc, t[0])]                                 # the step's maximum, log2 domain
        if not lazy:
            for ins in full: f(ins)
        if init:
            f(I("v_mov_b32", m, t[0]), I("v_xor_b32", negm, 0x80000000, t[0]), I("v_add_f32", mt, THR_BITS, t[0]))
            if self.lazy_max: f(I("v_add_f32", t[1], -0.5, mt), I("v_mul_f32", self.vMTR[qb], self.s_rsc, t[1]))
        else:
            st = self.new_site()
            if lazy:
                # trigger: the lane's PARTIAL maximum (its half of the row's keys), raw, against (m + THR - 0.5) / scale — fires whenever the exact
                # test below could (0.5 in the log2 domain covers the rounding of the two forms by orders of magnitude); a spurious trigger changes
                # nothing: the slow path repeats the exact per-lane test on the row's full maximum
                f(I("v_cmp_gt_f32", VCC, t[0], self.vMTR[qb]), I("s_cbranch_vccnz", f"L_slow_{st}", ndef=0, xuse=[VCC]), Label(f"L_slowret_{st}"))
            else:
                f(I("v_cmp_gt_f32", VCC, t[0], mt), I("s_cbranch_vccnz", f"L_slow_{st}", ndef=0, xuse=[VCC]), Label(f"L_slowret_{st}"))
            # slow path: lanes whose step maximum exceeds m + THR take it as the new m; l is rescaled now, O at the next step boundary
            a = t[2]
            self.ool += [Label(f"L_slow_{st}")] + ((full + [I("v_cmp_gt_f32", VCC, t[0], mt)]) if lazy else []) + [
                         I("v_cndmask_b32", t[1], m, t[0], VCC),                    # m_new = vcc ? ms : m
                         I("v_sub_f32", a, m, t[1]),
                         I("v_exp_f32", a, a),
                         I("v_mov_b32", m, t[1]),
                         I("v_xor_b32", negm, 0x80000000, t[1]),
                         I("v_add_f32", mt, THR_BITS, t[1]),
                         I("v_mul_f32", l, l, a),
                         I("v_mul_f32", self.vALPHA[qb], self.vALPHA[qb], a)] + ([I("v_add_f32", t[1], -0.5, mt), I("v_mul_f32", self.vMTR[qb], self.s_rsc, t[1])] if lazy else []) + [
                         I("s_bitset1_b32", self.s_pend, qb, rw=True),
                         I("s_branch", f"L_slowret_{st}", ndef=0)]
        # p = exp2(s * scale - m); row sum; pack to bf16
        consumers = []          # (needs: highest x index, Ins)
        if self.dot_sum:
            tail = [I("v_add_f32", acc[0], acc[0], acc[1]), I("v_add_f32", l, l, acc[0])]
        else:
            pairs = [(0, 1), (2, 3), (4, 5), (6, 7)]
            for k, (a_, b_) in enumerate(pairs): consumers.append((b_, I("v_add_f32", acc[k], x[a_], x[b_])))
            for i in range(8, 16): consumers.append((i, I("v_add_f32", acc[i & 3], acc[i & 3], x[i])))
            tail = [I("v_add_f32", acc[0], acc[0], acc[1]), I("v_add_f32", acc[2], acc[2], acc[3]), I("v_add_f32", acc[0], acc[0], acc[2]),
                    I("v_add_f32", l, l, acc[0])]
        order = sorted(range(8), key=lambda j: 8 * (j >> 2) + 2 * (j & 3) + 1)
        for n_, j in enumerate(order):
            sp, k = j >> 2, j & 3
            need = 8 * sp + 2 * k + 1
            consumers.append((need, I("v_cvt_pk_bf16_f32", P[j], x[8 * sp + 2 * k], x[8 * sp + 2 * k + 1])))
            if self.dot_sum:
                a_ = acc[n_ & 1]
                consumers.append((need + 0.5, I("v_dot2_f32_bf16", a_, P[j], self.s_ones, 0 if n_ < 2 else a_)))
        consumers.sort(key=lambda c: c[0])
        for i in range(16): f(I("v_fma_f32", x[i], x[i], sc, negm))
        ci = 0
        for i in range(16):
            f(I("v_exp_f32", x[i], x[i]))
            # consumers whose operands were exponentiated at least two instructions ago
            while ci < len(consumers) and consumers[ci][0] <= i - 2:
                f(consumers[ci][1]); ci += 1
                if ci % 2 == 0: break
        while ci < len(consumers): f(consumers[ci][1]); ci += 1
        for ins in tail: f(ins)
        return F

    def mask_block(self, qb, p, n_sgpr, tag):
        """Out-of-line: S[qb][p] entries whose key offset exceeds the limit become -inf.  Step s+1 (the one whose scores these
        are) is the block's diagonal step when s + 1 == n - 1 (limit = l31 - 4 hi) and past its end when s + 1 >= n (limit -1)."""
        st = self.new_site()
        x = self.vS[qb][p]
        blk = [Label(f"L_mask_{st}"),
               I("s_add_u32", self.s_t[0], self.s_step, 2, comment=tag),
               I("s_cmp_eq_u32", self.s_t[0], n_sgpr, ndef=0),                         # s + 2 == n  <=>  s + 1 == n - 1
               I("s_cbranch_scc1", f"L_mask_tri_{st}", ndef=0),
               I("v_mov_b32", self.vLIMC, -1),
               I("s_branch", f"L_mask_go_{st}", ndef=0),
               Label(f"L_mask_tri_{st}"),
               I("v_mov_b32", self.vLIMC, self.vLIM),
               Label(f"L_mask_go_{st}"),
               s_nop(11)]                                                              # (the scores come from the previous step's MFMAs)
        for i in range(16):
            c = (i & 3) + 8 * (i >> 2)
            blk += [I("v_cmp_gt_i32", VCC, c, self.vLIMC), I("v_cndmask_b32", x[i], x[i], self.vNINF, VCC)]
        blk += [I("s_branch", f"L_maskret_{st}", ndef=0)]
        self.ool += blk
        # in line: if (step + 2 >= n) goto mask
        return [I("s_add_u32", self.s_t[0], self.s_step, 2), I("s_cmp_ge_u32", self.s_t[0], n_sgpr, ndef=0),
                I("s_cbranch_scc1", f"L_mask_{st}", ndef=0), Label(f"L_maskret_{st}")]

    def rescale_check(self, qbs):
        """In line at the top of a step: apply pending O rescales (out of line)."""
        st = self.new_site()
        blk = [Label(f"L_resc_{st}"), s_nop(15)]
        for qb in qbs:
            blk += [I("s_bitcmp1_b32", self.s_pend, qb, ndef=0), I("s_cbranch_scc0", f"L_resc_{st}_skip{qb}", ndef=0)]
            tmp = self.vX
            for r0 in range(0, 64, 8):
                for k in range(8): blk.append(I("v_accvgpr_read_b32", tmp[k], self.aO[qb][r0 + k]))
                for k in range(8): blk.append(I("v_mul_f32", tmp[k], tmp[k], self.vALPHA[qb]))
                for k in range(8): blk.append(I("v_accvgpr_write_b32", self.aO[qb][r0 + k], tmp[k]))
            blk += [I("v_mov_b32", self.vALPHA[qb], 1.0), Label(f"L_resc_{st}_skip{qb}")]
        blk += [I("s_mov_b32", self.s_pend, 0), s_nop(1), I("s_branch", f"L_rescret_{st}", ndef=0)]
        self.ool += blk
        return [I("s_cmp_lg_u32", self.s_pend, 0, ndef=0), I("s_cbranch_scc1", f"L_resc_{st}", ndef=0), Label(f"L_rescret_{st}")]

    # ---- one step body ----------------------------------------------------------------------------------------------
    def body(self, mode, first):
        """mode 'D': both blocks, 'S': qb1 only.  first: odd step of an iteration (p = 1, fragment half 0, carries the iteration's
        8 DMA pieces as fillers); second: even step (p = 0, half 1)."""
        p = 1 if first else 0
        half = 0 if first else 1
        qbs = (0, 1) if mode == "D" else (1,)
        mf, k_after, v_after = [], {}, {}
        if mode == "D":
            for i in range(16):
                ks, qb = i >> 1, i & 1
                mf.append(self.qk(qb, p, ks)); k_after[ks] = len(mf) - 1
                sp, db, qb2 = i >> 3, (i >> 1) & 3, i & 1
                mf.append(self.pv(qb2, p, sp, db)); v_after[(sp, db)] = len(mf) - 1
        else:
            for i in range(8):
                mf.append(self.qk(1, p, i)); k_after[i] = len(mf) - 1
                sp, db = i >> 2, i & 3
                mf.append(self.pv(1, p, sp, db)); v_after[(sp, db)] = len(mf) - 1
        early = []
        # canonical LDS order K0, V0, K1, V1, ... (identical in every body: the lgkmcnt arithmetic relies on it)
        for j in range(8):
            if "nolds" in self.abl: break
            early.append(Filler([self.k_read(j, half)], after=k_after[j], prio=0))
            sp, db = j >> 2, j & 3
            early.append(Filler(self.v_reads(sp, db, half), after=v_after[(sp, db)], prio=0))
        if first and "nodma" not in self.abl:
            for gi, grp in enumerate(self.dma_stage_ins()):
                early.append(Filler(grp, after=gi, prio=0))
        stream = []
        n_ool = len(self.ool)
        # masks first (they branch out of line only on the last two steps of a block), then the softmax chains
        for qb in (qbs if mode == "S" else (0,)):
            n = self.s_n0 if qb == 0 else self.s_n1
            stream.append(Filler(self.mask_block(qb, 1 - p, n, f"mask qb{qb}"), cost=3))
        chains = [self.sm_stream(qb, 1 - p) for qb in qbs]
        if "nosm" in self.abl: chains, stream, self.ool = [[] for _ in qbs], [], self.ool[:n_ool]
        if len(chains) == 2:                      # interleave the two blocks' chains instruction by instruction (ILP)
            a, b = chains
            for i in range(max(len(a), len(b))):
                if i < len(a): stream.append(a[i])
                if i < len(b): stream.append(b[i])
        else:
            stream += chains[0]
        if "nomfma" in self.abl: mf = [s_nop(0) for _ in mf]
        return self.rescale_check(qbs) + interleave(mf, early, stream, head_room=2)

    def finish_body(self, items, tag):
        # the entry state of every body: the 24 reloads of the previous step in canonical order, all possibly outstanding
        canon = []
        for j in range(8):
            canon.append(self.aK.sub(4 * j, 4).ids())
            base = self.aV.sub(4 * j, 4)
            canon += [base.sub(0, 2).ids(), base.sub(2, 2).ids()]
        items, pend = add_lgkm_waits(items, initial=canon)
        tail = [x for x in items if isinstance(x, Ins)][-20:]
        items = fix_hazards(items, tag, self.report, carry=tail)
        if self.debug_serial:
            out = []
            for x in items:
                out.append(x)
                if isinstance(x, Ins) and (x.is_mfma or x.is_lds or x.is_valu): out += [s_waitcnt(lgkm=0), s_nop(15)]
            items = out
        return items

    # ---- item parameters from a queue index ------------------------------------------------------------------------------
    def emit_params(self, P, li, nxt):
        """li (queue index inside this XCD's chunk) -> (pair, block) -> descriptors and offsets.  nxt = False: this item's full set;
        True: the subset the prefetch needs (s_rn, s_kcoln, s_vcoln, s_rb0n, s_rb1n, s_hqn).  Clobbers s_t[0..7], s_t64, lgkmcnt."""
        add = P.add
        t = self.s_t
        st = self.new_site()
        bh, jj, b, h, j, S_, tok0 = t[0], t[1], t[2], t[3], t[4], t[5], t[6]
        # groups of 8 pairs, heaviest block of every pair first: li = g * 8 nqb + jj * 8 + bi  (the last group may hold fewer pairs)
        add(I("s_cmp_lt_u32", li, self.s_split, ndef=0)); add(I("s_cbranch_scc0", f"L_par_short_{st}", ndef=0))
        add(I("s_lshr_b32", t[7], li, 3)); add(I("s_and_b32", bh, li, 7))
        add(I("s_cmp_eq_u32", self.s_nqb, 1, ndef=0)); add(I("s_cbranch_scc1", f"L_par_nqb1_{st}", ndef=0))
        add(I("s_mul_hi_u32", t[2], t[7], self.s_magq)); add(I("s_mul_i32", t[3], t[2], self.s_nqb)); add(I("s_sub_u32", jj, t[7], t[3]))
        add(I("s_branch", f"L_par_g_{st}", ndef=0))
        P.label(f"L_par_nqb1_{st}"); add(I("s_mov_b32", t[2], t[7])); add(I("s_mov_b32", jj, 0))
        P.label(f"L_par_g_{st}")
        add(I("s_lshl_b32", t[2], t[2], 3)); add(I("s_add_u32", bh, bh, t[2]))
        add(I("s_branch", f"L_par_bh_{st}", ndef=0))
        P.label(f"L_par_short_{st}")                                   # r = li - split; jj = r / rem; bi = r % rem  (rem in 1..7)
        add(I("s_sub_u32", bh, li, self.s_split)); add(I("s_mov_b32", jj, 0))
        P.label(f"L_par_div_{st}")
        add(I("s_cmp_lt_u32", bh, self.s_rem, ndef=0)); add(I("s_cbranch_scc1", f"L_par_divd_{st}", ndef=0))
        add(I("s_sub_u32", bh, bh, self.s_rem)); add(I("s_add_u32", jj, jj, 1)); add(I("s_branch", f"L_par_div_{st}", ndef=0))
        P.label(f"L_par_divd_{st}")
        add(I("s_lshr_b32", t[2], self.s_split, 3)); add(I("s_cmp_eq_u32", self.s_nqb, 1, ndef=0)); add(I("s_cbranch_scc1", f"L_par_s1_{st}", ndef=0))
        add(I("s_mul_hi_u32", t[2], t[2], self.s_magq))                                    # full groups = split / (8 nqb)
        P.label(f"L_par_s1_{st}")
        add(I("s_lshl_b32", t[2], t[2], 3)); add(I("s_add_u32", bh, bh, t[2]))
        P.label(f"L_par_bh_{st}")
        add(I("s_add_u32", bh, bh, self.s_bh0))
        add(I("s_sub_u32", j, self.s_nqb, jj)); add(I("s_sub_u32", j, j, 1))                 # heaviest block first
        add(I("s_cmp_eq_u32", self.s_H, 1, ndef=0)); add(I("s_cbranch_scc1", f"L_par_h1_{st}", ndef=0))
        add(I("s_mul_hi_u32", b, bh, self.s_magh)); add(I("s_mul_i32", t[7], b, self.s_H)); add(I("s_sub_u32", h, bh, t[7]))
        add(I("s_branch", f"L_par_hd_{st}", ndef=0))
        P.label(f"L_par_h1_{st}"); add(I("s_mov_b32", b, bh)); add(I("s_mov_b32", h, 0))
        P.label(f"L_par_hd_{st}")
        # sequence: padded [B, S_max] or packed (cu_seqlens)
        add(I("s_cmp_eq_u64", self.s_cu, 0, ndef=0)); add(I("s_cbranch_scc1", f"L_par_pad_{st}", ndef=0))
        add(I("s_lshl_b32", t[7], b, 2)); add(I("s_load_dwordx2", self.s_t64, self.s_cu, t[7])); add(s_waitcnt(lgkm=0))
        add(I("s_mov_b32", tok0, self.s_t64[0])); add(I("s_sub_u32", S_, self.s_t64[1], self.s_t64[0]))
        add(I("s_branch", f"L_par_seq_{st}", ndef=0))
        P.label(f"L_par_pad_{st}"); add(I("s_mul_i32", tok0, b, self.s_smax)); add(I("s_mov_b32", S_, self.s_smax))
        P.label(f"L_par_seq_{st}")
        rq = self.s_rn if nxt else self.s_rq
        add(I("s_mul_i32", t[7], tok0, self.s_ld2)); add(I("s_mul_hi_u32", t[0], tok0, self.s_ld2))           # (bh is dead from here)
        add(I("s_add_u32", rq[0], self.s_qkv[0], t[7])); add(I("s_addc_u32", rq[1], self.s_qkv[1], t[0]))
        add(I("s_and_b32", rq[1], rq[1], 0xffff))
        add(I("s_mul_i32", rq[2], S_, self.s_ld2)); add(I("s_mov_b32", rq[3], 0x00020000))
        if not nxt:
            add(I("s_mov_b32", self.s_numrec, rq[2])); add(I("s_mov_b32", self.s_S, S_)); add(I("s_mov_b32", self.s_h, h)); add(I("s_mov_b32", self.s_j, j))
            add(I("s_mul_i32", t[7], tok0, self.s_outrow)); add(I("s_mul_hi_u32", t[0], tok0, self.s_outrow))
            add(I("s_add_u32", self.s_ro[0], self.s_out[0], t[7])); add(I("s_addc_u32", self.s_ro[1], self.s_out[1], t[0]))
            add(I("s_and_b32", self.s_ro[1], self.s_ro[1], 0xffff))
            add(I("s_mul_i32", self.s_ro[2], S_, self.s_outrow)); add(I("s_mov_b32", self.s_ro[3], 0x00020000))
            add(I("s_mul_i32", t[7], b, self.s_H)); add(I("s_add_u32", t[7], t[7], h)); add(I("s_mul_i32", self.s_lsebase, t[7], self.s_smax))
        # K / V column offsets of this wave's DMA pieces
        kcol, vcol = (self.s_kcoln, self.s_vcoln) if nxt else (self.s_kcol, self.s_vcol)
        add(I("s_add_u32", t[7], self.s_H, h)); add(I("s_lshl_b32", t[7], t[7], 8))
        add(I("s_lshr_b32", t[0], self.s_w, 1)); add(I("s_lshl_b32", t[1], self.s_ld2, 5)); add(I("s_mul_i32", t[0], t[0], t[1]))
        add(I("s_and_b32", t[1], self.s_w, 1)); add(I("s_lshl_b32", t[1], t[1], 7))
        if self.k_rows:
            add(I("s_mov_b32", kcol, t[7]))                                                   # (the wave's rows are in vVOFF)
        else:
            add(I("s_add_u32", kcol, t[7], t[0])); add(I("s_add_u32", kcol, kcol, t[1]))
        add(I("s_lshl_b32", t[7], self.s_H, 1)); add(I("s_add_u32", t[7], t[7], h)); add(I("s_lshl_b32", vcol, t[7], 8))
        if nxt: add(I("s_lshl_b32", self.s_hqn, h, 8))
        # this wave's blocks: bA = 8 j + w, bB = bA + 4
        rb0, rb1 = (self.s_rb0n, self.s_rb1n) if nxt else (self.s_rb0, self.s_rb1)
        add(I("s_add_u32", t[0], S_, 31)); add(I("s_lshr_b32", t[0], t[0], 5))                             # nblk
        add(I("s_lshl_b32", t[1], j, 3)); add(I("s_add_u32", t[2], t[1], self.s_w)); add(I("s_add_u32", t[3], t[2], 4))
        add(I("s_lshl_b32", rb0, t[2], 5)); add(I("s_lshl_b32", t[7], t[3], 5))
        if nxt:
            add(I("s_cmp_lt_u32", t[3], t[0], ndef=0)); add(I("s_cselect_b32", rb1, t[7], rb0))
        else:
            add(I("s_add_u32", t[5], t[2], 1)); add(I("s_add_u32", t[6], t[3], 1))                         # bA + 1, bB + 1  (S_, tok0 are dead)
            add(I("s_cmp_lt_u32", t[2], t[0], ndef=0)); add(I("s_cselect_b32", t[4], t[5], 0))             # validA ? bA + 1 : 0  (j is dead)
            add(I("s_cmp_lt_u32", t[3], t[0], ndef=0))
            add(I("s_cselect_b32", self.s_n1, t[6], t[4])); add(I("s_cselect_b32", self.s_n0, t[5], 0))
            add(I("s_cselect_b32", rb1, t[7], rb0))
            # key tiles of the item: ceil(min(8 j + 8, nblk) / 2), none if the block lies past the sequence
            add(I("s_add_u32", t[2], t[1], 8)); add(I("s_min_u32", t[2], t[2], t[0])); add(I("s_add_u32", t[2], t[2], 1)); add(I("s_lshr_b32", t[2], t[2], 1))
            add(I("s_cmp_lt_u32", t[1], t[0], ndef=0)); add(I("s_cselect_b32", self.s_nT, t[2], 0))

    def q_prefetch(self, rsrc, rb0, rb1, hq):
        """16 loads: the Q rows of both blocks -> vQN."""
        X = self.vX
        out = []
        for qb, rb in ((0, rb0), (1, rb1)):
            out += [I("v_add_u32", X[0], rb, self.vL31), I("v_mul_lo_u32", X[0], X[0], self.s_ld2),
                    I("v_lshlrev_b32", X[1], 4, self.vHI), I("v_add3_u32", X[2 + qb], X[0], X[1], hq)]
            for ks in range(8):
                out.append(I("buffer_load_dwordx4", self.vQN.sub(32 * qb + 4 * ks, 4), X[2 + qb], rsrc, 0, mods=f"offen offset:{32 * ks}"))
        return out

    def queue_pop(self):
        """wave 0, lane 0: vATOM = fetch_add(queue[xcd], 1) (device scope); consumed one item later."""
        st = self.new_site()
        return [I("s_cmp_eq_u32", self.s_w, 0, ndef=0), I("s_cbranch_scc0", f"L_pop_{st}", ndef=0),
                I("s_mov_b64", EXEC, 1), I("v_mov_b32", self.vX[0], 0), I("v_mov_b32", self.vX[1], 1),
                I("global_atomic_add", self.vATOM, self.vX[0], self.vX[1], self.s_qx, mods="sc0 sc1"),
                I("s_mov_b64", EXEC, -1), Label(f"L_pop_{st}")]

    def stamp(self, k):
        """prof: add the cycles since the previous stamp to sum k."""
        if not self.prof: return []
        return [s_waitcnt(lgkm=0), I("s_memtime", self.s_tm), s_waitcnt(lgkm=0),
                I("s_sub_u32", self.s_t[7], self.s_tm[0], self.s_t0), I("s_mov_b32", self.s_t0, self.s_tm[0]),
                I("s_mov_b64", EXEC, 1 << k), I("v_add_u32", self.vPACC, self.s_t[7], self.vPACC), I("s_mov_b64", EXEC, -1)]

    # ---- the whole kernel ----------------------------------------------------------------------------------------------
    def build(self):
        P = Program()
        t = self.s_t
        add = P.add
        X = self.vX
        # ------------------------------------------------------------------ arguments, per-lane constants
        add(I("s_load_dwordx8", self.s_args0, self.s_karg, 0x0))
        add(I("s_load_dwordx8", self.s_args1, self.s_karg, 0x20))
        add(I("s_load_dwordx4", self.s_args2, self.s_karg, 0x40))
        if self.prof:
            add(I("s_load_dwordx2", self.s_prof, self.s_karg, 0x50))
            add(I("v_mov_b32", self.vPACC, 0))
        add(I("v_and_b32", self.vLANE, 63, self.v_tid))
        add(I("v_lshrrev_b32", X[0], 6, self.v_tid))
        add(I("v_and_b32", self.vL31, 31, self.vLANE))
        add(I("v_lshrrev_b32", self.vHI, 5, self.vLANE))
        add(s_nop(0))
        add(I("v_readfirstlane_b32", self.s_w, X[0]))
        add(s_waitcnt(lgkm=0))
        if self.lazy_max:                          # 1 / scale_log2 (the arguments have landed)
            add(I("v_rcp_f32", self.s_rsc, self.s_scale))
        if self.prof:
            add(I("s_memtime", self.s_tm)); add(s_waitcnt(lgkm=0)); add(I("s_mov_b32", self.s_t0, self.s_tm[0]))
        add(I("s_lshl_b32", self.s_ld2, self.s_ld, 1)); add(I("s_lshl_b32", self.s_outrow, self.s_H, 8))
        add(I("s_lshl_b32", self.s_tile, self.s_ld2, 6)); add(I("s_lshl_b32", self.s_r16, self.s_ld2, 4))
        add(I("v_mov_b32", self.vNINF, NEG_INF))
        if self.dot_sum: add(I("v_mov_b32", self.s_ones, 0x3F803F80))
        add(I("v_lshlrev_b32", X[0], 2, self.vHI)); add(I("v_sub_u32", self.vLIM, self.vL31, X[0]))
        add(I("v_lshlrev_b32", self.vKADDR, 4, self.vLANE))
        if self.k_rows:          # tile_addr<128>(l31, 2 ks + hi) = l31 * 256 + (((2 ks + hi) ^ swz(l31)) << 4), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3)
            add(I("v_and_b32", X[0], 3, self.vL31)); add(I("v_bfe_u32", X[1], self.vL31, 2, 2)); add(I("v_lshl_or_b32", X[0], X[0], 2, X[1]))
            for ks in range(8):
                add(I("v_or_b32", X[1], 2 * ks, self.vHI)); add(I("v_xor_b32", X[1], X[1], X[0])); add(I("v_lshlrev_b32", X[1], 4, X[1]))
                add(I("v_lshl_add_u32", self.vKADDRS[ks], self.vL31, 8, X[1]))
        # V tr-read addresses (attention.hip v_off): p16 = l & 15, g16 = (l & 31) >> 4
        p16, g16, r, c, sw = X[1], X[2], X[3], X[4], X[5]
        add(I("v_and_b32", p16, 15, self.vLANE)); add(I("v_lshrrev_b32", g16, 4, self.vL31))
        add(I("v_lshrrev_b32", X[6], 2, p16))                                                   # p16 >> 2
        add(I("v_lshl_add_u32", r, self.vHI, 2, X[6]))                                          # r = 4 hi + (p16 >> 2)
        add(I("v_lshl_or_b32", sw, X[6], 2, self.vHI))                                          # swz = ((p16>>2) << 2) | hi
        add(I("v_bfe_u32", X[7], p16, 1, 1))                                                    # (p16 & 3) >> 1
        add(I("v_lshl_or_b32", c, g16, 1, X[7]))                                                # c = 2 g16 + ...
        add(I("v_xor_b32", c, c, sw)); add(I("v_lshlrev_b32", c, 4, c))
        add(I("v_lshl_add_u32", c, r, 8, c))                                                    # r * 256 + ((c ^ swz) << 4)
        add(I("v_and_b32", X[7], 1, p16)); add(I("v_lshl_add_u32", c, X[7], 3, c))              # + (p16 & 1) * 8
        for db in range(4):
            add(I("v_xor_b32", self.vVADDR[db][0], db << 6, c))
            add(I("v_xor_b32", X[8], (db << 6) ^ 32, c)); add(I("v_add_u32", self.vVADDR[db][1], 2048, X[8]))
        # DMA source offsets
        add(I("v_mul_lo_u32", X[0], self.vL31, self.s_ld2)); add(I("v_lshl_add_u32", self.vKOFF, self.vHI, 4, X[0]))
        add(I("v_lshrrev_b32", X[0], 4, self.vLANE))                                            # lane >> 4
        add(I("v_lshl_add_u32", X[1], self.s_w, 2, X[0]))                                       # 4 w + (lane >> 4)
        add(I("v_mul_lo_u32", X[1], X[1], self.s_ld2))
        add(I("v_lshl_or_b32", X[2], X[0], 2, self.s_w))                                        # swz = ((lane >> 4) << 2) | w
        add(I("v_and_b32", X[3], 15, self.vLANE)); add(I("v_xor_b32", X[3], X[3], X[2]))
        add(I("v_lshl_add_u32", self.vVOFF, X[3], 4, X[1]))
        # ------------------------------------------------------------------ this XCD's queue
        # workgroup g runs on XCD g % 8 (observed, for L2 locality only: correctness rests on the queue counter being device scope);
        # XCD x owns the pairs [bh0, bh0 + nbh) and its workgroups pop queue[x]
        add(I("s_and_b32", t[0], self.s_wg, 7)); add(I("s_lshr_b32", t[1], self.s_wg, 3))                  # x, c
        add(I("s_lshr_b32", t[2], self.s_G, 3)); add(I("s_and_b32", t[3], self.s_G, 7))
        add(I("s_cmp_lt_u32", t[0], t[3], ndef=0)); add(I("s_cselect_b32", t[3], 1, 0)); add(I("s_add_u32", self.s_nwg, t[2], t[3]))
        add(I("s_lshr_b32", t[2], self.s_BH, 3)); add(I("s_and_b32", t[3], self.s_BH, 7))
        add(I("s_mul_i32", t[4], t[0], t[2])); add(I("s_min_u32", t[5], t[0], t[3])); add(I("s_add_u32", self.s_bh0, t[4], t[5]))
        add(I("s_cmp_lt_u32", t[0], t[3], ndef=0)); add(I("s_cselect_b32", t[3], 1, 0)); add(I("s_add_u32", t[2], t[2], t[3]))   # nbh
        add(I("s_mul_i32", self.s_items, t[2], self.s_nqb))
        add(I("s_and_b32", self.s_rem, t[2], 7)); add(I("s_lshr_b32", t[3], t[2], 3)); add(I("s_lshl_b32", t[3], t[3], 3))
        add(I("s_mul_i32", self.s_split, t[3], self.s_nqb))
        add(I("s_lshl_b32", t[4], t[0], 6)); add(I("s_add_u32", self.s_qx[0], self.s_queue[0], t[4])); add(I("s_addc_u32", self.s_qx[1], self.s_queue[1], 0))
        add(I("s_cmp_ge_u32", t[1], self.s_items, ndef=0)); add(I("s_cbranch_scc1", "L_exit", ndef=0))
        add(I("s_mov_b32", self.s_li, t[1]))
        add(self.queue_pop())                                                                   # the index of the second item
        add(I("s_mov_b32", self.s_bc, 0)); add(I("s_mov_b32", self.s_cs, 0))
        self.emit_params(P, self.s_li, nxt=False)
        # first item: nothing was prefetched by a predecessor -> first K tile, stages 0..2, Q now
        add(I("s_mov_b32", self.s_rd[0], self.s_rq[0])); add(I("s_mov_b32", self.s_rd[1], self.s_rq[1]))
        add(I("s_mov_b32", self.s_rd[2], self.s_rq[2])); add(I("s_mov_b32", self.s_rd[3], self.s_rq[3]))
        add(self.dma_k0_ins(self.s_kcol, self.s_rd))
        add(I("s_add_u32", self.s_ktile, self.s_kcol, self.s_tile)); add(I("s_mov_b32", self.s_vtile, self.s_vcol))
        add(I("s_mov_b32", t[5], 0)); add(self.dma_set_slot(t[5]))
        for _ in range(3):
            for g in self.dma_stage_ins(): add(g)
            add(self.dma_advance())
        add(I("s_lshl_b32", t[5], self.s_h, 8))
        add(self.q_prefetch(self.s_rq, self.s_rb0, self.s_rb1, t[5]))
        # ================================================================== item
        P.label("L_item")
        add(s_waitcnt(vm=0))                              # this item's Q (vQN), first K tile and stages (this wave's pieces), the queue pop
        st = self.new_site()
        add(I("s_cmp_eq_u32", self.s_w, 0, ndef=0)); add(I("s_cbranch_scc0", f"L_bc_{st}", ndef=0))
        add(s_nop(0)); add(I("v_readfirstlane_b32", t[0], self.vATOM)); add(I("s_add_u32", t[0], t[0], self.s_nwg))
        add(I("v_mov_b32", X[0], t[0])); add(I("s_add_u32", t[1], self.s_bc, BCAST)); add(I("v_mov_b32", X[1], t[1]))
        add(I("ds_write_b32", X[1], X[0], ndef=0))
        P.label(f"L_bc_{st}")
        for qb in range(2):
            for i in range(32): add(I("v_accvgpr_write_b32", self.aQ[qb][i], self.vQN[32 * qb + i]))
        for qb in range(2):
            for i in range(64): add(I("v_accvgpr_write_b32", self.aO[qb][i], 0))
            add(I("v_mov_b32", self.vL[qb], 0)); add(I("v_mov_b32", self.vALPHA[qb], 1.0))
        add(I("s_mov_b32", self.s_pend, 0)); add(I("s_mov_b32", self.s_step, 0)); add(I("s_mov_b32", self.s_U, 0))
        add(I("s_mov_b32", self.s_cs0, self.s_cs))
        add(s_waitcnt(lgkm=0)); add(I("s_barrier", ndef=0))
        add(self.stamp(0))
        # the next item: queue index from wave 0, then the parameters its prefetch needs
        add(I("s_add_u32", t[1], self.s_bc, BCAST)); add(I("v_mov_b32", X[1], t[1])); add(I("ds_read_b32", X[0], X[1])); add(s_waitcnt(lgkm=0))
        add(I("v_readfirstlane_b32", self.s_lin, X[0])); add(I("s_xor_b32", self.s_bc, self.s_bc, 4))
        add(I("s_cmp_lt_u32", self.s_lin, self.s_items, ndef=0)); add(I("s_cbranch_scc1", "L_next_valid", ndef=0))
        for i in range(4): add(I("s_mov_b32", self.s_rn[i], self.s_rq[i]))
        add(I("s_mov_b32", self.s_rn[2], 0))                                                    # no next item: its loads fetch nothing
        add(I("s_mov_b32", self.s_kcoln, 0)); add(I("s_mov_b32", self.s_vcoln, 0)); add(I("s_mov_b32", self.s_rb0n, 0)); add(I("s_mov_b32", self.s_rb1n, 0))
        add(I("s_mov_b32", self.s_hqn, 0))
        add(I("s_branch", "L_next_done", ndef=0))
        P.label("L_next_valid")
        self.emit_params(P, self.s_lin, nxt=True)
        P.label("L_next_done")
        # the DMA stream continues with this item's stage 3 (stages 0..2 were issued by the predecessor)
        for i in range(4): add(I("s_mov_b32", self.s_rd[i], self.s_rq[i]))
        add(I("s_lshl_b32", t[0], self.s_tile, 2)); add(I("s_add_u32", self.s_ktile, self.s_kcol, t[0]))
        add(I("s_sub_u32", t[0], t[0], self.s_tile)); add(I("s_add_u32", self.s_vtile, self.s_vcol, t[0]))
        add(self.dma_set_slot(self.s_cs))
        add(I("s_mov_b32", self.s_dso, 3)); add(I("s_mov_b32", self.s_nu, 0))
        # idle wave (no valid block): it still serves the barriers and its share of the DMA
        add(I("s_cmp_eq_u32", self.s_n1, 0, ndef=0)); add(I("s_cbranch_scc1", "L_after_step0", ndef=0))
        # QK of steps 0 and 1 from the first K tile, softmax of step 0, fragments for step 0's body
        pro = []
        if self.k_rows:
            # addresses relative to the K0 area: the row-image registers minus the ring position (s_cs * SLOT)
            add(I("s_lshl_b32", t[0], self.s_cs, 15)); add(I("s_sub_u32", t[0], K0AREA, t[0]))
            k0 = [self.vQN[ks] for ks in range(8)]                               # (vQN is free here: Q was moved to the AccVGPRs above)
            for ks in range(8): add(I("v_add_u32", k0[ks], t[0], self.vKADDRS[ks]))
        else:
            add(I("v_lshlrev_b32", X[9], 4, self.vLANE)); add(I("v_add_u32", X[9], K0AREA, X[9]))
            k0 = X[9]
        for half in range(2):
            for ks in range(8): pro.append(self.k_read(ks, half, addr=k0))
            for ks in range(8):
                for qb in range(2): pro.append(self.qk(qb, half, ks))
        for ks in range(8): pro.append(self.k_read(ks, 0))                                      # K tile 1 half 0 -> QK of step 2
        for jx in range(8): pro += self.v_reads(jx >> 2, jx & 3, 0)                             # V tile 0 half 0 -> PV of step 0
        pro, _ = add_lgkm_waits(pro)
        add(pro)
        add(s_nop(15))
        for qb in range(2):
            n = self.s_n0 if qb == 0 else self.s_n1
            st = self.new_site()
            add(I("s_cmp_eq_u32", n, 0, ndef=0)); add(I("s_cbranch_scc1", f"L_pro_skip_{st}", ndef=0))
            add(I("s_cmp_eq_u32", n, 1, ndef=0)); add(I("s_cbranch_scc0", f"L_pro_nomask_{st}", ndef=0))
            add(I("v_mov_b32", self.vLIMC, self.vLIM))
            for i in range(16):
                c_ = (i & 3) + 8 * (i >> 2)
                add(I("v_cmp_gt_i32", VCC, c_, self.vLIMC)); add(I("v_cndmask_b32", self.vS[qb][0][i], self.vS[qb][0][i], self.vNINF, VCC))
            P.label(f"L_pro_nomask_{st}")
            for fl in self.sm_stream(qb, 0, init=True): add(fl.ins)
            P.label(f"L_pro_skip_{st}")
        add(s_waitcnt(lgkm=0))
        # step 0 = a "second" body
        add(I("s_cmp_lg_u32", self.s_n0, 0, ndef=0)); add(I("s_cbranch_scc0", "L_step0_S", ndef=0))
        add(self.finish_body(self.body("D", first=False), "step0_D"))
        add(I("s_branch", "L_after_step0", ndef=0))
        P.label("L_step0_S")
        add(self.finish_body(self.body("S", first=False), "step0_S"))
        P.label("L_after_step0")
        add(self.stamp(1))
        add(I("s_cmp_eq_u32", self.s_nT, 0, ndef=0)); add(I("s_cbranch_scc1", "L_flush", ndef=0))
        add(self.consumer_advance())
        # ------------------------------------------------------------------ main loop: iteration U = steps 2U+1, 2U+2 on stage U+1
        P.label("L_loop")
        add(s_waitcnt(vm=8))                              # everything but the 8 pieces of the previous iteration has landed
        add(I("s_barrier", ndef=0))
        # what the DMA stream issues in this iteration: the item's next stage, or a unit of the NEXT item
        add(I("s_cmp_lt_u32", self.s_dso, self.s_nT, ndef=0)); add(I("s_cbranch_scc0", "L_duty_next", ndef=0))
        add(I("s_add_u32", self.s_dso, self.s_dso, 1)); add(I("s_branch", "L_duty_done", ndef=0))
        P.label("L_duty_next")
        add(self.next_unit_prepare("loop"))
        P.label("L_duty_done")
        add(I("s_lshl_b32", t[1], self.s_U, 1)); add(I("s_add_u32", self.s_step, t[1], 1))
        for mode, lbl in (("D", "L_first_S"), ("S", "L_first_idle")):
            n = self.s_n0 if mode == "D" else self.s_n1
            add(I("s_cmp_lt_u32", self.s_step, n, ndef=0)); add(I("s_cbranch_scc0", lbl, ndef=0))
            add(self.finish_body(self.body(mode, first=True), f"first_{mode}"))
            add(I("s_branch", "L_second", ndef=0))
            P.label(lbl)
        if "nodma" not in self.abl:
            for g in self.dma_stage_ins(): add(g)                                                # idle wave: DMA only
        P.label("L_second")
        add(self.dma_advance())
        add(I("s_add_u32", self.s_step, self.s_step, 1))
        for mode, lbl in (("D", "L_second_S"), ("S", "L_second_idle")):
            n = self.s_n0 if mode == "D" else self.s_n1
            add(I("s_cmp_lt_u32", self.s_step, n, ndef=0)); add(I("s_cbranch_scc0", lbl, ndef=0))
            add(self.finish_body(self.body(mode, first=False), f"second_{mode}"))
            add(I("s_branch", "L_iter_end", ndef=0))
            P.label(lbl)
        P.label("L_iter_end")
        add(I("s_add_u32", self.s_U, self.s_U, 1))
        add(I("s_cmp_ge_u32", self.s_U, self.s_nT, ndef=0)); add(I("s_cbranch_scc1", "L_flush", ndef=0))
        add(self.consumer_advance())                      # (not after the last iteration: the ring position of the next item's stage 0)
        add(I("s_branch", "L_loop", ndef=0))
        # ------------------------------------------------------------------ what is left of the next item's units (short items)
        P.label("L_flush")
        add(self.stamp(2))
        add(I("s_cmp_ge_u32", self.s_nu, 3, ndef=0)); add(I("s_cbranch_scc1", "L_epilogue", ndef=0))
        add(self.next_unit_prepare("flush"))
        for g in self.dma_stage_ins(): add(g)
        add(self.dma_advance())
        add(I("s_branch", "L_flush", ndef=0))
        # ------------------------------------------------------------------ epilogue, next item
        P.label("L_epilogue")
        add(s_nop(15))
        add(s_waitcnt(lgkm=0))
        for qb in range(2): add(self.epilogue(qb))
        add(self.stamp(3))
        add(I("s_cmp_lt_u32", self.s_lin, self.s_items, ndef=0)); add(I("s_cbranch_scc0", "L_done", ndef=0))
        add(I("s_mov_b32", self.s_li, self.s_lin))
        self.emit_params(P, self.s_li, nxt=False)
        add(self.stamp(4))
        add(I("s_branch", "L_item", ndef=0))
        P.label("L_done")
        if self.prof:
            add(s_waitcnt(vm=0))
            add(I("s_cmp_eq_u64", self.s_prof, 0, ndef=0)); add(I("s_cbranch_scc1", "L_exit", ndef=0))
            # 8 dwords per wave at prof + (wg * 4 + w) * 32: lane k < 5 stores cycle sum k
            add(I("s_lshl_b32", t[4], self.s_wg, 2)); add(I("s_add_u32", t[4], t[4], self.s_w)); add(I("s_lshl_b32", t[4], t[4], 5))
            add(I("v_lshlrev_b32", X[8], 2, self.vLANE)); add(I("v_add_u32", X[8], t[4], X[8]))
            add(I("s_mov_b64", EXEC, 0xff))
            add(I("global_store_dword", X[8], self.vPACC, self.s_prof, ndef=0))
            add(I("s_mov_b64", EXEC, -1))
        P.label("L_exit")
        add(s_waitcnt(vm=0))
        add(I("s_endpgm", ndef=0))
        add(self.ool)
        return P

    def next_unit_prepare(self, tag):
        """Set the DMA stream up for unit s_nu of the NEXT item (0: first K tile + stage 0, 1: stage 1, 2: stage 2 + Q rows + the
        queue pop for the item after), then count it.  The 8 stage pieces themselves are issued by the caller."""
        t = self.s_t
        st = self.new_site()
        out = [I("s_cmp_eq_u32", self.s_nu, 0, ndef=0), I("s_cbranch_scc0", f"L_nu_not0_{st}", ndef=0)]
        for i in range(4): out.append(I("s_mov_b32", self.s_rd[i], self.s_rn[i]))
        # ring slot of the next item's stage 0 = (slot at this item's start + nT) mod 3
        out += [I("s_add_u32", t[5], self.s_cs0, self.s_nT), I("s_mul_hi_u32", t[4], t[5], 0x55555556), I("s_mul_i32", t[4], t[4], 3),
                I("s_sub_u32", t[5], t[5], t[4])]
        out += self.dma_set_slot(t[5])
        out += self.dma_k0_ins(self.s_kcoln, self.s_rd)
        out += [I("s_add_u32", self.s_ktile, self.s_kcoln, self.s_tile), I("s_mov_b32", self.s_vtile, self.s_vcoln)]
        out += [Label(f"L_nu_not0_{st}"), I("s_cmp_eq_u32", self.s_nu, 2, ndef=0), I("s_cbranch_scc0", f"L_nu_not2_{st}", ndef=0)]
        out += self.q_prefetch(self.s_rn, self.s_rb0n, self.s_rb1n, self.s_hqn)
        out += self.queue_pop()
        out += [Label(f"L_nu_not2_{st}"), I("s_add_u32", self.s_nu, self.s_nu, 1)]
        return out

    def epilogue(self, qb):
        st = self.new_site()
        X = self.vX
        n = self.s_n0 if qb == 0 else self.s_n1
        rb = self.s_rb0 if qb == 0 else self.s_rb1
        t = self.s_t
        out = [I("s_cmp_eq_u32", n, 0, ndef=0), I("s_cbranch_scc1", f"L_epi_skip_{st}", ndef=0)]
        l, m = self.vL[qb], self.vM[qb]
        out += [I("v_mov_b32", X[0], l), s_nop(1), I("v_permlane32_swap_b32", X[0], l, ndef=2), I("v_add_f32", l, l, X[0]),
                I("v_rcp_f32", X[1], l), I("v_log_f32", X[2], l), s_nop(0),
                I("v_add_f32", X[2], X[2], m), I("v_mul_f32", X[2], LN2, X[2])]
        # lse[(b*H + h)*S_max + row] for row < S, from the lanes with hi == 0
        out += [I("s_cmp_eq_u64", self.s_lse, 0, ndef=0), I("s_cbranch_scc1", f"L_epi_nolse_{st}", ndef=0),
                I("v_add_u32", X[3], rb, self.vL31),
                I("v_cmp_gt_u32", VCC, self.s_S, X[3]),
                I("v_cmp_eq_u32", self.s_m64, 0, self.vHI),
                I("s_and_b64", self.s_m64, self.s_m64, VCC),
                I("v_add_u32", X[3], self.s_lsebase, X[3]), I("v_lshlrev_b32", X[3], 2, X[3]),
                I("s_and_saveexec_b64", self.s_exec, self.s_m64, xdef=[EXEC]),
                I("global_store_dword", X[3], X[2], self.s_lse, ndef=0),
                I("s_mov_b64", EXEC, self.s_exec),
                Label(f"L_epi_nolse_{st}")]
        # O * (1/l) -> bf16 -> wave-private LDS block [32 rows][256 B], chunk c of row r at c ^ (r & 15)
        out += [I("v_lshlrev_b32", X[4], 8, self.vL31), I("v_lshl_add_u32", X[4], self.vHI, 3, X[4]),
                I("v_and_b32", X[5], 15, self.vL31), I("v_lshl_add_u32", X[4], X[5], 4, X[4]),
                I("s_lshl_b32", t[0], self.s_w, 13), I("s_add_u32", t[0], t[0], OSTAGE), I("v_add_u32", X[4], t[0], X[4])]
        for db in range(4):
            for g in range(4):
                c = 4 * db + g
                src = self.aO[qb].sub(16 * db + 4 * g, 4)
                out += [I("v_accvgpr_read_b32", X[8 + k], src[k]) for k in range(4)]
                out += [I("v_mul_f32", X[8 + k], X[8 + k], X[1]) for k in range(4)]
                out += [I("v_cvt_pk_bf16_f32", X[6], X[8], X[9]), I("v_cvt_pk_bf16_f32", X[7], X[10], X[11]),
                        I("v_xor_b32", X[5], c << 4, X[4]),
                        I("ds_write_b64", X[5], X[6:8], ndef=0)]
        out += [s_waitcnt(lgkm=0)]
        # rows back out: instruction `it` covers rows 4 it + (lane >> 4), 16 B per lane; all eight reads, then all eight stores
        out += [I("v_lshrrev_b32", X[0], 4, self.vLANE), I("v_and_b32", X[3], 15, self.vLANE),
                I("v_xor_b32", X[5], X[3], X[0]), I("v_lshlrev_b32", X[5], 4, X[5]), I("v_lshl_add_u32", X[5], X[0], 8, X[5]),
                I("v_add_u32", X[5], t[0], X[5]),
                # global: (rb + 4 it + (lane >> 4)) * H*256 + h*256 + (lane & 15) * 16
                I("v_mul_lo_u32", X[2], X[0], self.s_outrow), I("v_lshl_add_u32", X[2], X[3], 4, X[2]),
                I("s_mul_i32", t[1], rb, self.s_outrow), I("s_lshl_b32", t[2], self.s_h, 8), I("s_add_u32", t[1], t[1], t[2]),
                I("s_lshl_b32", t[3], self.s_outrow, 2)]
        buf = self.vS[0][0], self.vS[0][1]                    # (the score registers are free here: 8 x 4 registers of row data)
        regs = [buf[it >> 2].sub(4 * (it & 3), 4) for it in range(8)]
        for it in range(8):
            out += [I("v_xor_b32", X[4], (it & 3) << 6, X[5]), I("ds_read_b128", regs[it], X[4], mods=f"offset:{1024 * it}")]
        for it in range(8):
            out += [s_waitcnt(lgkm=7 - it),
                    I("buffer_store_dwordx4", regs[it], X[2], self.s_ro, t[1], ndef=0, mods="offen"),
                    I("s_add_u32", t[1], t[1], t[3])]
        out += [s_nop(1), Label(f"L_epi_skip_{st}")]
        return fix_hazards(out, f"epilogue{qb}", self.report)


def generate(debug_serial=False, abl=(), prof=False, name=KNAME, k_rows=True, dot_sum=False, lazy_max=False):
    g = Gen(debug_serial=debug_serial, abl=abl, prof=prof, k_rows=k_rows, dot_sum=dot_sum, lazy_max=lazy_max)
    prog = g.build()
    items = fix_hazards(prog.items, "whole", g.report)
    body = render(items)
    hdr = ("// GENERATED by csrc/asm/attn_fwd_d128.py — do not edit.  Register map:\n" + g.S.listing() + "\n" + g.V.listing() + "\n"
           + g.A.listing() + "\n// hazard pads inserted: " + ", ".join(f"{t}={n}" for t, n in g.report))
    n_mfma = sum(1 for x in items if isinstance(x, Ins) and x.is_mfma)
    args = ARGS + ([(8, 80, "global_buffer")] if prof else [])
    text = kernel_text(name, body, lds_bytes=LDS_BYTES, kernarg_bytes=KERNARG_BYTES + (8 if prof else 0), n_arch_vgpr=g.V.next,
                       n_acc_vgpr=256, n_sgpr=g.S.next + 2, args_meta=args, header_comment=hdr)
    return text, g, n_mfma


# development variants (make DEV=1): cycle sums and timing-only ablations (hvlm_attn_fwd_asm_dev)
DEV_VARIANTS = [("prof", dict(prof=True)), ("nosm", dict(prof=True, abl=("nosm",))), ("nodma", dict(prof=True, abl=("nodma",))),
                ("nolds", dict(prof=True, abl=("nolds",))), ("nomfma", dict(prof=True, abl=("nomfma",))),
                ("mfmaonly", dict(prof=True, abl=("nosm", "nodma", "nolds"))), ("kgran", dict(prof=True, k_rows=False)), ("dotsum", dict(prof=True, dot_sum=True)), ("kgrandot", dict(prof=True, dot_sum=True, k_rows=False)), ("lazymax", dict(prof=True, lazy_max=True))]


def unique_labels(text, suffix):
    return re.sub(r"\bL_([A-Za-z0-9_]+)", lambda m: f"L{suffix}_{m.group(1)}", text)


if __name__ == "__main__":
    text, g, n = generate(debug_serial="--serial" in sys.argv)
    if "--dev" in sys.argv:
        # one code object with the production kernel and every development variant (labels made unique per kernel)
        parts = [text]
        for i, (vn, kw) in enumerate(DEV_VARIANTS):
            vt, _, _ = generate(name=f"{KNAME}_{vn}", **kw)
            vt = unique_labels(vt, f"v{i}")
            vt = "\n".join(l for l in vt.split("\n") if not l.startswith(".amdgcn_target") and not l.startswith(".amdhsa_code_object_version"))
            parts.append(vt)
        text = "\n".join(parts)
    out = [a for a in sys.argv[1:] if not a.startswith("--")]
    if out:
        with open(out[0], "w") as f: f.write(text)
    else:
        sys.stdout.write(text)
    sys.stderr.write(f"attn_fwd_d128: {n} MFMA, arch VGPRs {g.V.next}, SGPRs {g.S.next}; pads: {g.report}\n")
