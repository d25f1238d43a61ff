#!/usr/bin/env python3
"""Causal flash-attention FORWARD for dh = 128, bf16, gfx950 — hand-allocated registers, one wave per SIMD.

Replaces, for the decoder's shapes, attn_fwd_kernel<128, true, 4> of attention.hip (same operands, same results to rounding):
    softmax(Q K^T / sqrt(dh) + causal mask) V          (transformers LlamaAttention; the flash path of the reference,
                                                        llava/train/llama_flash_attn_monkey_patch.py:81-94, incl. cu_seqlens)
reading q / k / v straight from the token-major fused buffer qkv[tokens, ld] and writing O[tokens, H*128] and lse[B, H, S_max].

Structure (why it is not the C++ kernel re-typed):
  * workgroup = 4 waves = ONE wave per SIMD, 512 registers each; a workgroup owns 256 query rows = eight 32-row blocks and wave w
    owns blocks w and w + 4 ("qb0", "qb1"): near the diagonal every wave still has work, so the lock-step K/V stream wastes
    2 of (16 j + 12) half-units instead of a whole wave (DESIGN.md "attention, round 5" has the count);
  * per wave:  O (2 x 64) , Q (2 x 32), K fragments (32) and V fragments (32) live in AccVGPRs — exactly 256 — and the scores,
    probabilities and softmax state in ~140 arch VGPRs;
  * a step = 32 keys.  Three-stage software pipeline with NO dependency inside a step:
        MFMA stream :  QK^T of step s+2  (-> S[p])   alternating with   P V of step s  (<- P[p])
        VALU stream :  softmax of step s+1  (S[1-p] -> P[1-p], running max / sum)
        LDS stream  :  K fragments of step s+3 and V fragments of step s+1, each reloaded right after its last use
    so the matrix pipe never waits for the exponentials of its own tile (what bounded the C++ kernels: r03d_valu_mfma_probe);
  * the running maximum is only RAISED when some score exceeds it by 2^THR (THR = 8): the O rescale (128 AccVGPRs through the
    VALU) is an out-of-line slow path, applied at the next step boundary when every P V of the old scale has been issued;
  * K / V tiles (64 keys) arrive by LDS-DMA into a two-stage ring, stage X = {K tile X+1, V tile X}: one barrier per tile,
    the next stage's DMA issued as MFMA fillers right behind it.  K's LDS image is [half][ks][hi][key] so that a fragment read
    is base + immediate (one address register); V keeps the XOR-swizzled row-major image that ds_read_b64_tr_b16 needs.

The file is a GENERATOR: `python3 attn_fwd_d128.py out.s` writes the assembly (asmgen.py pads hazards and derives the lgkmcnt
waits); the Makefile assembles it into a code object that attention_asm.hip embeds and launches through hipModuleLaunchKernel.
"""
import sys
from asmgen import (Reg, RegFile, Program, I, Ins, Label, Raw, Filler, VCC, M0, EXEC, s_nop, s_waitcnt, mfma_bf16, interleave,
                    fix_hazards, add_lgkm_waits, render, kernel_text)

KNAME = "hvlm_fa_fwd_d128_causal"
THR_BITS = 0x41000000          # 8.0f: the running maximum is raised only when a score exceeds it by more than 2^8
NEG_INF = 0xff800000
LN2 = 0x3f317218

# LDS map -------------------------------------------------------------------------------------------------------------
SLOT = 32768                   # stage = K tile (16 KiB) + V tile (16 KiB)
V_IN_SLOT = 16384
STAGING = 65536                # 4 x 8 KiB wave-private blocks for the O epilogue; K tile 0 sits here during the prologue
LDS_BYTES = STAGING + 4 * 8192

# kernel arguments ------------------------------------------------------------------------------------------------------
ARGS = [(8, 0, "global_buffer"), (8, 8, "global_buffer"), (8, 16, "global_buffer"), (8, 24, "global_buffer"),
        (4, 32, "by_value"), (4, 36, "by_value"), (4, 40, "by_value"), (4, 44, "by_value"),
        (4, 48, "by_value"), (4, 52, "by_value"), (4, 56, "by_value"), (4, 60, "by_value")]
KERNARG_BYTES = 64


class Gen:
    def __init__(self, debug_serial=False, abl=(), prof=False):
        """abl: timing-only ablations (WRONG RESULTS): 'nosm' no softmax stream, 'nodma' no LDS-DMA in the loop, 'nolds' no
        fragment reloads, 'nomfma' no MFMAs, 'noqk' / 'nopv' one MFMA family only.  prof: s_memtime stamps (entry, after the
        prologue loads, loop entry, loop exit, end) stored per wave through the extra kernel argument."""
        self.debug_serial = debug_serial
        self.abl, self.prof = set(abl), prof
        self.report = []
        self.site = 0
        self.ool = []                     # out-of-line blocks (slow paths), appended behind the main code
        S = self.S = RegFile("s", 0, 100)
        V = self.V = RegFile("v", 0, 256)
        A = self.A = RegFile("a", 0, 256)
        # ---- SGPRs ------------------------------------------------------------------------------------------------
        self.s_karg = S.alloc("kernarg", 2)
        self.s_wg = S.alloc("workgroup id", 1)
        S.alloc("pad", 1)
        self.s_rq = S.alloc("rsrc qkv (sample rows)", 4, 4)
        self.s_args0 = S.alloc("args 0..7: qkv, out, lse, cu", 8, 4)
        self.s_args1 = S.alloc("args 8..15: S_max H ld scale nqb total magic_nqb magic_H", 8, 4)
        self.s_ro = S.alloc("rsrc out (sample rows)", 4, 4)
        self.s_rd = S.alloc("rsrc for DMA (num_records zeroed past the last tile)", 4, 4)
        self.s_qkv, self.s_out, self.s_lse, self.s_cu = (self.s_args0.sub(0, 2), self.s_args0.sub(2, 2), self.s_args0.sub(4, 2),
                                                         self.s_args0.sub(6, 2))
        (self.s_smax, self.s_H, self.s_ld, self.s_scale, self.s_nqb, self.s_total, self.s_magq, self.s_magh) = [self.s_args1[i] for i in range(8)]
        self.s_t = [S.alloc(f"tmp{i}", 1) for i in range(8)]
        self.s_t64 = S.alloc("tmp64", 2, 2)
        self.s_exec = S.alloc("saved exec", 2, 2)
        self.s_m64 = S.alloc("mask64", 2, 2)
        self.s_b = S.alloc("b", 1); self.s_h = S.alloc("h", 1); self.s_j = S.alloc("j (256-row block)", 1)
        self.s_S = S.alloc("S (this sample)", 1); self.s_tok0 = S.alloc("tok0", 1)
        self.s_w = S.alloc("wave", 1)
        self.s_n0 = S.alloc("n0: steps of qb0", 1); self.s_n1 = S.alloc("n1: steps of qb1", 1)
        self.s_rb0 = S.alloc("first row of qb0", 1); self.s_rb1 = S.alloc("first row of qb1", 1)
        self.s_nT = S.alloc("key tiles of the workgroup", 1)
        self.s_U = S.alloc("iteration", 1); self.s_step = S.alloc("step", 1)
        self.s_pend = S.alloc("pending O rescale bits", 1)
        self.s_ld2 = S.alloc("ld * 2 bytes", 1)
        self.s_kcol = S.alloc("K soffset base of this wave", 1); self.s_vcol = S.alloc("V soffset base", 1)
        self.s_tile = S.alloc("64 * ld * 2", 1); self.s_r16 = S.alloc("16 * ld * 2", 1)
        self.s_ktile = S.alloc("soffset of the next K tile", 1); self.s_vtile = S.alloc("soffset of the next V tile", 1)
        self.s_m0k = S.alloc("m0 base K pieces", 1); self.s_m0v = S.alloc("m0 base V pieces", 1)
        self.s_numrec = S.alloc("num_records qkv", 1)
        self.s_so = S.alloc("soffset tmp", 1)
        self.s_lsebase = S.alloc("(b*H+h)*S_max", 1)
        self.s_outrow = S.alloc("H*256", 1)
        self.s_prof = S.alloc("prof buffer", 2, 2)
        self.s_stamp = [S.alloc(f"stamp{i}", 2, 2) for i in range(5)]
        self.s_it = S.alloc("iteration stamp", 2, 2)
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
        self.vL = [V.alloc(f"l qb{q}", 1) for q in range(2)]
        self.vALPHA = [V.alloc(f"pending alpha qb{q}", 1) for q in range(2)]
        self.vT = [V.alloc(f"softmax temps qb{q}", 6) for q in range(2)]
        self.vACC = [V.alloc(f"partial sums qb{q}", 4) for q in range(2)]
        self.vNINF = V.alloc("-inf", 1)
        self.vLIM = V.alloc("l31 - 4 hi (triangular limit)", 1)
        self.vLIMC = V.alloc("limit in use", 1)
        self.vKADDR = V.alloc("K fragment LDS address (lane*16 + slot)", 1)
        self.vVADDR = [[V.alloc(f"V tr-read address db{db} hf{hf}", 1) for hf in range(2)] for db in range(4)]
        self.vKOFF = V.alloc("K DMA voffset", 1)
        self.vVOFF = V.alloc("V DMA voffset", 1)
        self.vLANE = V.alloc("lane", 1); self.vL31 = V.alloc("lane & 31", 1); self.vHI = V.alloc("lane >> 5", 1)
        self.vX = V.alloc("scratch", 12, 4)
        self.vPROF = [V.alloc(f"iteration stamps {i} (lane = iteration)", 1) for i in range(4)]

    # ---------------------------------------------------------------------------------------------------------------
    def new_site(self):
        self.site += 1
        return self.site

    # ---- DMA of one stage (K tile X+1, V tile X) by this wave: 8 pieces -------------------------------------------------
    def dma_stage_ins(self, slot_sgpr_k, slot_sgpr_v, ktile, vtile, rsrc):
        """8 groups of (m0, soffset, buffer_load lds).  slot_sgpr_k / _v: SGPRs holding the LDS byte address of this wave's
        first K / V piece in the target slot."""
        g = []
        for i in range(4):
            g.append([I("s_add_u32", M0, slot_sgpr_k, 0x400 * i) if i else I("s_mov_b32", M0, slot_sgpr_k),
                      I("s_add_u32", self.s_so, ktile, 32 * i) if i else I("s_mov_b32", self.s_so, ktile),
                      I("buffer_load_dwordx4", self.vKOFF, rsrc, self.s_so, ndef=0, mods="offen lds", xuse=[M0])])
        for i in range(4):
            g.append([I("s_add_u32", M0, slot_sgpr_v, 0x1000 * i) if i else I("s_mov_b32", M0, slot_sgpr_v),
                      I("s_add_u32", self.s_so, self.s_so, self.s_r16) if i else I("s_mov_b32", self.s_so, vtile),
                      I("buffer_load_dwordx4", self.vVOFF, rsrc, self.s_so, ndef=0, mods="offen lds", xuse=[M0])])
        return g

    # ---- fragment reads -------------------------------------------------------------------------------------------
    def k_read(self, ks, half, addr=None, extra=0):
        return I("ds_read_b128", self.aK.sub(4 * ks, 4), addr or self.vKADDR, mods=f"offset:{extra + half * 8192 + ks * 1024}")

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
        """returns (list of Filler in order, out-of-line slow path)"""
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
        f(I("v_mov_b32", t[1], t[0]))
        # (the hazard pass pads VALU -> permlane; with two qb chains interleaved the other chain's instructions sit in between)
        f(I("v_permlane32_swap_b32", t[1], t[0], ndef=2))
        f(I("v_max_f32", t[0], t[0], t[1]))
        f(I("v_mul_f32", t[0], sc, t[0]))                                   # the step's maximum, log2 domain
        if init:
            f(I("v_mov_b32", m, t[0]), I("v_xor_b32", negm, 0x80000000, t[0]), I("v_add_f32", mt, THR_BITS, t[0]))
        else:
            st = self.new_site()
            f(I("v_cmp_gt_f32", VCC, t[0], mt), I("s_cbranch_vccnz", f"L_slow_{st}", ndef=0, xuse=[VCC]), Label(f"L_slowret_{st}"))
            # slow path: lanes whose step maximum exceeds m + THR take it as the new m; l is rescaled now, O at the next step boundary
            a = t[2]
            self.ool += [Label(f"L_slow_{st}"),
                         I("v_cndmask_b32", t[1], m, t[0], VCC),                    # m_new = vcc ? ms : m
                         I("v_sub_f32", a, m, t[1]),
                         I("v_exp_f32", a, a),
                         I("v_mov_b32", m, t[1]),
                         I("v_xor_b32", negm, 0x80000000, t[1]),
                         I("v_add_f32", mt, THR_BITS, t[1]),
                         I("v_mul_f32", l, l, a),
                         I("v_mul_f32", self.vALPHA[qb], self.vALPHA[qb], a),
                         I("s_bitset1_b32", self.s_pend, qb, rw=True),
                         I("s_branch", f"L_slowret_{st}", ndef=0)]
        # p = exp2(s * scale - m); row sum; pack to bf16
        consumers = []          # (needs: highest x index, Ins)
        pairs = [(0, 1), (2, 3), (4, 5), (6, 7)]
        for k, (a_, b_) in enumerate(pairs): consumers.append((b_, I("v_add_f32", acc[k], x[a_], x[b_])))
        for i in range(8, 16): consumers.append((i, I("v_add_f32", acc[i & 3], acc[i & 3], x[i])))
        tail = [I("v_add_f32", acc[0], acc[0], acc[1]), I("v_add_f32", acc[2], acc[2], acc[3]), I("v_add_f32", acc[0], acc[0], acc[2]),
                I("v_add_f32", l, l, acc[0])]
        for j in range(8):
            sp, k = j >> 2, j & 3
            consumers.append((8 * sp + 2 * k + 1, I("v_cvt_pk_bf16_f32", P[j], x[8 * sp + 2 * k], x[8 * sp + 2 * k + 1])))
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
        """mode 'D': both blocks, 'S': qb1 only.  first: odd step of an iteration (p = 1, fragment half 0, carries the DMA
        issue of the next stage); second: even step (p = 0, half 1, toggles the LDS slot of the address registers at its end)."""
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
            groups = self.dma_stage_ins(self.s_m0k, self.s_m0v, self.s_ktile, self.s_vtile, self.s_rd)
            if "nodmak" in self.abl: groups = groups[4:]
            if "nodmav" in self.abl: groups = groups[:4]
            for gi, grp in enumerate(groups):
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
        if not first:
            tog = [I("v_xor_b32", self.vKADDR, 0x8000, self.vKADDR)]
            for db in range(4):
                for hf in range(2): tog.append(I("v_xor_b32", self.vVADDR[db][hf], 0x8000, self.vVADDR[db][hf]))
            # after the last LDS read of the step
            stream.append(Filler(tog, after=len(mf) - 1))
        if "nomfma" in self.abl: mf = [s_nop(0) for _ in mf]
        if "noqk" in self.abl: mf = [m if (isinstance(m, Ins) and m.ops[0].kind == "a") else s_nop(0) for m in mf]
        if "nopv" in self.abl: mf = [m if (isinstance(m, Ins) and m.ops[0].kind == "v") else s_nop(0) for m in mf]
        items = self.rescale_check(qbs) + interleave(mf, early, stream, head_room=2)
        if not first:
            # the toggles must follow every ds_read of the body: move them to the very end
            togs = [x for x in items if isinstance(x, Ins) and x.op == "v_xor_b32" and x.ops[1] == 0x8000]
            items = [x for x in items if not (isinstance(x, Ins) and x.op == "v_xor_b32" and x.ops[1] == 0x8000)] + togs
        return items

    # ---- the whole kernel ----------------------------------------------------------------------------------------------
    def build(self):
        P = Program()
        s, v = self, self
        t = self.s_t
        sa0, sa1 = self.s_args0, self.s_args1
        add = P.add
        # ------------------------------------------------------------------ arguments, mapping
        add(I("s_load_dwordx8", sa0, self.s_karg, 0x0))
        add(I("s_load_dwordx8", sa1, self.s_karg, 0x20))
        if self.prof:
            add(I("s_load_dwordx2", self.s_prof, self.s_karg, 0x40))
            add(I("s_memtime", self.s_stamp[0]))
            for k in range(4): add(I("v_mov_b32", self.vPROF[k], 0))
        add(I("v_and_b32", self.vLANE, 63, self.v_tid))
        add(I("v_lshrrev_b32", self.vX[0], 6, self.v_tid))
        add(I("v_and_b32", self.vL31, 31, self.vLANE))
        add(I("v_lshrrev_b32", self.vHI, 5, self.vLANE))
        add(s_nop(0))
        add(I("v_readfirstlane_b32", self.s_w, self.vX[0]))
        add(s_waitcnt(lgkm=0))
        # XCD-aware order: workgroup id -> logical id (consecutive logical ids on one XCD)
        add(I("s_and_b32", t[0], self.s_wg, 7)); add(I("s_lshr_b32", t[1], self.s_total, 3)); add(I("s_and_b32", t[2], self.s_total, 7))
        add(I("s_mul_i32", t[3], t[0], t[1])); add(I("s_min_u32", t[4], t[0], t[2])); add(I("s_add_u32", t[3], t[3], t[4]))
        add(I("s_lshr_b32", t[4], self.s_wg, 3)); add(I("s_add_u32", t[3], t[3], t[4]))             # t3 = logical id
        # bh = id / nqb, jj = id % nqb
        add(I("s_cmp_eq_u32", self.s_nqb, 1, ndef=0)); add(I("s_cbranch_scc1", "L_nqb1", ndef=0))
        add(I("s_mul_hi_u32", t[4], t[3], self.s_magq)); add(I("s_mul_i32", t[5], t[4], self.s_nqb)); add(I("s_sub_u32", t[5], t[3], t[5]))
        add(I("s_branch", "L_nqb_done", ndef=0))
        P.label("L_nqb1"); add(I("s_mov_b32", t[4], t[3])); add(I("s_mov_b32", t[5], 0))
        P.label("L_nqb_done")
        add(I("s_sub_u32", self.s_j, self.s_nqb, t[5])); add(I("s_sub_u32", self.s_j, self.s_j, 1))  # heaviest block first
        add(I("s_cmp_eq_u32", self.s_H, 1, ndef=0)); add(I("s_cbranch_scc1", "L_h1", ndef=0))
        add(I("s_mul_hi_u32", self.s_b, t[4], self.s_magh)); add(I("s_mul_i32", t[6], self.s_b, self.s_H)); add(I("s_sub_u32", self.s_h, t[4], t[6]))
        add(I("s_branch", "L_h_done", ndef=0))
        P.label("L_h1"); add(I("s_mov_b32", self.s_b, t[4])); add(I("s_mov_b32", self.s_h, 0))
        P.label("L_h_done")
        # sequence: padded [B, S_max] or packed (cu_seqlens)
        add(I("s_cmp_eq_u64", self.s_cu, 0, ndef=0)); add(I("s_cbranch_scc1", "L_padded", ndef=0))
        add(I("s_lshl_b32", t[0], self.s_b, 2)); add(I("s_load_dwordx2", self.s_t64, self.s_cu, t[0])); add(s_waitcnt(lgkm=0))
        add(I("s_mov_b32", self.s_tok0, self.s_t64[0])); add(I("s_sub_u32", self.s_S, self.s_t64[1], self.s_t64[0]))
        add(I("s_branch", "L_seq_done", ndef=0))
        P.label("L_padded"); add(I("s_mul_i32", self.s_tok0, self.s_b, self.s_smax)); add(I("s_mov_b32", self.s_S, self.s_smax))
        P.label("L_seq_done")
        add(I("s_lshl_b32", t[0], self.s_j, 8)); add(I("s_cmp_ge_u32", t[0], self.s_S, ndef=0)); add(I("s_cbranch_scc1", "L_exit", ndef=0))
        # descriptors over this sample's rows
        add(I("s_lshl_b32", self.s_ld2, self.s_ld, 1))
        add(I("s_mul_i32", t[0], self.s_tok0, self.s_ld2)); add(I("s_mul_hi_u32", t[1], self.s_tok0, self.s_ld2))
        add(I("s_add_u32", self.s_rq[0], self.s_qkv[0], t[0])); add(I("s_addc_u32", self.s_rq[1], self.s_qkv[1], t[1]))
        add(I("s_and_b32", self.s_rq[1], self.s_rq[1], 0xffff))
        add(I("s_mul_i32", self.s_numrec, self.s_S, self.s_ld2)); add(I("s_mov_b32", self.s_rq[2], self.s_numrec)); add(I("s_mov_b32", self.s_rq[3], 0x00020000))
        add(I("s_mov_b32", self.s_rd[0], self.s_rq[0])); add(I("s_mov_b32", self.s_rd[1], self.s_rq[1]))
        add(I("s_mov_b32", self.s_rd[2], self.s_numrec)); add(I("s_mov_b32", self.s_rd[3], 0x00020000))
        add(I("s_lshl_b32", self.s_outrow, self.s_H, 8))
        add(I("s_mul_i32", t[0], self.s_tok0, self.s_outrow)); add(I("s_mul_hi_u32", t[1], self.s_tok0, self.s_outrow))
        add(I("s_add_u32", self.s_ro[0], self.s_out[0], t[0])); add(I("s_addc_u32", self.s_ro[1], self.s_out[1], t[1]))
        add(I("s_and_b32", self.s_ro[1], self.s_ro[1], 0xffff))
        add(I("s_mul_i32", self.s_ro[2], self.s_S, self.s_outrow)); add(I("s_mov_b32", self.s_ro[3], 0x00020000))
        add(I("s_mul_i32", t[0], self.s_b, self.s_H)); add(I("s_add_u32", t[0], t[0], self.s_h)); add(I("s_mul_i32", self.s_lsebase, t[0], self.s_smax))
        # this wave's blocks
        add(I("s_add_u32", t[0], self.s_S, 31)); add(I("s_lshr_b32", t[0], t[0], 5))                      # nblk
        add(I("s_lshl_b32", t[1], self.s_j, 3)); add(I("s_add_u32", t[2], t[1], self.s_w)); add(I("s_add_u32", t[3], t[2], 4))   # bA, bB
        add(I("s_lshl_b32", self.s_rb0, t[2], 5))
        add(I("s_add_u32", t[4], t[2], 1)); add(I("s_add_u32", t[5], t[3], 1))                             # bA + 1, bB + 1
        add(I("s_lshl_b32", t[7], t[3], 5))                                                            # (SALU shifts / adds clobber SCC:
        add(I("s_cmp_lt_u32", t[2], t[0], ndef=0)); add(I("s_cselect_b32", t[6], t[4], 0))                   #  compare and selects stay adjacent)
        add(I("s_cmp_lt_u32", t[3], t[0], ndef=0))
        add(I("s_cselect_b32", self.s_n1, t[5], t[6])); add(I("s_cselect_b32", self.s_n0, t[4], 0))
        add(I("s_cselect_b32", self.s_rb1, t[7], self.s_rb0))
        add(I("s_add_u32", t[1], t[1], 8)); add(I("s_min_u32", t[1], t[1], t[0])); add(I("s_add_u32", t[1], t[1], 1)); add(I("s_lshr_b32", self.s_nT, t[1], 1))
        # DMA constants
        add(I("s_lshl_b32", self.s_tile, self.s_ld2, 6)); add(I("s_lshl_b32", self.s_r16, self.s_ld2, 4))
        add(I("s_add_u32", t[0], self.s_H, self.s_h)); add(I("s_lshl_b32", t[0], t[0], 8))                  # K column bytes
        add(I("s_lshr_b32", t[1], self.s_w, 1)); add(I("s_lshl_b32", t[2], self.s_ld2, 5)); add(I("s_mul_i32", t[1], t[1], t[2]))
        add(I("s_and_b32", t[2], self.s_w, 1)); add(I("s_lshl_b32", t[2], t[2], 7))                         # 4 (w & 1) * 32
        add(I("s_add_u32", self.s_kcol, t[0], t[1])); add(I("s_add_u32", self.s_kcol, self.s_kcol, t[2]))
        add(I("s_lshl_b32", t[0], self.s_H, 1)); add(I("s_add_u32", t[0], t[0], self.s_h)); add(I("s_lshl_b32", self.s_vcol, t[0], 8))
        # per-lane constants
        X = self.vX
        add(I("v_mov_b32", self.vNINF, NEG_INF))
        add(I("v_lshlrev_b32", X[0], 2, self.vHI)); add(I("v_sub_u32", self.vLIM, self.vL31, X[0]))
        add(I("v_lshlrev_b32", self.vKADDR, 4, self.vLANE))
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
        # ------------------------------------------------------------------ prologue DMA: K tile 0 -> staging, stages 0 and 1
        add(I("s_lshl_b32", t[0], self.s_w, 12))                                                # 4 w * 1024
        add(I("s_lshl_b32", t[1], self.s_w, 10))
        add(I("s_mov_b32", self.s_ktile, self.s_kcol)); add(I("s_mov_b32", self.s_vtile, self.s_vcol))
        add(I("s_add_u32", self.s_m0k, t[0], STAGING)); add(I("s_mov_b32", self.s_m0v, 0))
        k0 = self.dma_stage_ins(self.s_m0k, self.s_m0v, self.s_ktile, self.s_vtile, self.s_rd)[:4]          # K tile 0 only
        for g in k0: add(g)
        add(I("s_add_u32", self.s_ktile, self.s_ktile, self.s_tile))                            # stage 0 = K1, V0 -> slot 0
        add(I("s_mov_b32", self.s_m0k, t[0])); add(I("s_add_u32", self.s_m0v, t[1], V_IN_SLOT))
        for g in self.dma_stage_ins(self.s_m0k, self.s_m0v, self.s_ktile, self.s_vtile, self.s_rd): add(g)
        add(I("s_add_u32", self.s_ktile, self.s_ktile, self.s_tile)); add(I("s_add_u32", self.s_vtile, self.s_vtile, self.s_tile))
        add(I("s_add_u32", self.s_m0k, self.s_m0k, SLOT)); add(I("s_add_u32", self.s_m0v, self.s_m0v, SLOT))     # stage 1 -> slot 1
        for g in self.dma_stage_ins(self.s_m0k, self.s_m0v, self.s_ktile, self.s_vtile, self.s_rd): add(g)
        add(I("s_add_u32", self.s_ktile, self.s_ktile, self.s_tile)); add(I("s_add_u32", self.s_vtile, self.s_vtile, self.s_tile))
        add(I("s_sub_u32", self.s_m0k, self.s_m0k, SLOT)); add(I("s_sub_u32", self.s_m0v, self.s_m0v, SLOT))     # next: stage 2 -> slot 0
        # Q fragments straight into AccVGPRs
        for qb in range(2):
            rb = self.s_rb0 if qb == 0 else self.s_rb1
            add(I("v_add_u32", X[0], rb, self.vL31)); add(I("v_mul_lo_u32", X[0], X[0], self.s_ld2))
            add(I("s_lshl_b32", t[2], self.s_h, 8))
            add(I("v_lshlrev_b32", X[1], 4, self.vHI)); add(I("v_add3_u32", X[2 + qb], X[0], X[1], t[2]))
            for ks in range(8):
                add(I("buffer_load_dwordx4", self.aQ[qb].sub(4 * ks, 4), X[2 + qb], self.s_rq, 0, mods=f"offen offset:{32 * ks}"))
        # O = 0, softmax state
        for qb in range(2):
            for i in range(64): add(I("v_accvgpr_write_b32", self.aO[qb][i], 0))
            add(I("v_mov_b32", self.vL[qb], 0)); add(I("v_mov_b32", self.vALPHA[qb], 1.0))
        add(I("s_mov_b32", self.s_pend, 0)); add(I("s_mov_b32", self.s_step, 0)); add(I("s_mov_b32", self.s_U, 0))
        add(s_waitcnt(vm=0)); add(I("s_barrier", ndef=0))
        if self.prof: add(I("s_memtime", self.s_stamp[1])); add(s_waitcnt(lgkm=0))
        # idle wave (no valid block): it still serves the barriers and its share of the DMA
        add(I("s_cmp_eq_u32", self.s_n1, 0, ndef=0)); add(I("s_cbranch_scc1", "L_loop", ndef=0))
        # QK of steps 0 and 1 from K tile 0 (staging), softmax of step 0, fragments for step 0's body
        pro = []
        add(I("v_add_u32", X[9], STAGING, self.vKADDR))
        for half in range(2):
            for ks in range(8): pro.append(self.k_read(ks, half, addr=X[9]))
            for ks in range(8):
                for qb in range(2): pro.append(self.qk(qb, half, ks))
        for ks in range(8): pro.append(self.k_read(ks, 0))                                      # K tile 1 half 0 -> QK of step 2
        for j in range(8): pro += self.v_reads(j >> 2, j & 3, 0)                                # V tile 0 half 0 -> PV of step 0
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
        # step 0 = a "second" body on slot 0
        add(I("s_cmp_lg_u32", self.s_n0, 0, ndef=0)); add(I("s_cbranch_scc0", "L_step0_S", ndef=0))
        self.bodies = {}
        b = self.finish_body(self.body("D", first=False), "step0_D"); add(b)
        add(I("s_branch", "L_loop_entry" if self.prof else "L_loop", ndef=0))
        P.label("L_step0_S")
        b = self.finish_body(self.body("S", first=False), "step0_S"); add(b)
        # ------------------------------------------------------------------ main loop: iteration U = steps 2U+1, 2U+2 on stage U+1
        if self.prof:
            add(I("s_branch", "L_loop_entry", ndef=0))
        P.label("L_loop")
        if self.prof:
            add(I("s_branch", "L_loop2", ndef=0))
            P.label("L_loop_entry")
            add(s_waitcnt(lgkm=0)); add(I("s_memtime", self.s_stamp[2])); add(s_waitcnt(lgkm=0))
            P.label("L_loop2")
        add(I("s_cmp_ge_u32", self.s_U, self.s_nT, ndef=0)); add(I("s_cbranch_scc1", "L_epilogue", ndef=0))
        def itstamp(k):
            if self.prof:
                add(s_waitcnt(lgkm=0)); add(I("s_memtime", self.s_it)); add(s_waitcnt(lgkm=0))
                add(I("s_mov_b32", M0, self.s_U)); add(I("v_writelane_b32", self.vPROF[k], self.s_it[0], M0, rw=True))
        itstamp(0)
        add(s_waitcnt(vm=0)); add(I("s_barrier", ndef=0))
        itstamp(1)
        # the stage issued now is U + 2 (K tile U+3, V tile U+2): nothing to fetch once U + 2 >= nT
        add(I("s_add_u32", t[0], self.s_U, 2)); add(I("s_cmp_lt_u32", t[0], self.s_nT, ndef=0)); add(I("s_cselect_b32", self.s_rd[2], self.s_numrec, 0))
        add(I("s_lshl_b32", t[1], self.s_U, 1)); add(I("s_add_u32", self.s_step, t[1], 1))
        for mode, lbl in (("D", "L_first_S"), ("S", "L_first_idle")):
            n = self.s_n0 if mode == "D" else self.s_n1
            add(I("s_cmp_lt_u32", self.s_step, n, ndef=0)); add(I("s_cbranch_scc0", lbl, ndef=0))
            add(self.finish_body(self.body(mode, first=True), f"first_{mode}"))
            add(I("s_branch", "L_second", ndef=0))
            P.label(lbl)
        for g in self.dma_stage_ins(self.s_m0k, self.s_m0v, self.s_ktile, self.s_vtile, self.s_rd): add(g)   # idle wave: DMA only
        P.label("L_second")
        itstamp(2)
        add(I("s_add_u32", self.s_ktile, self.s_ktile, self.s_tile)); add(I("s_add_u32", self.s_vtile, self.s_vtile, self.s_tile))
        add(I("s_xor_b32", self.s_m0k, self.s_m0k, SLOT)); add(I("s_xor_b32", self.s_m0v, self.s_m0v, SLOT))
        add(I("s_add_u32", self.s_step, self.s_step, 1))
        for mode, lbl in (("D", "L_second_S"), ("S", "L_second_idle")):
            n = self.s_n0 if mode == "D" else self.s_n1
            add(I("s_cmp_lt_u32", self.s_step, n, ndef=0)); add(I("s_cbranch_scc0", lbl, ndef=0))
            add(self.finish_body(self.body(mode, first=False), f"second_{mode}"))
            add(I("s_branch", "L_iter_end", ndef=0))
            P.label(lbl)
        P.label("L_iter_end")
        itstamp(3)
        add(I("s_add_u32", self.s_U, self.s_U, 1)); add(I("s_branch", "L_loop", ndef=0))
        # ------------------------------------------------------------------ epilogue
        P.label("L_epilogue")
        add(s_nop(15))
        add(s_waitcnt(vm=0, lgkm=0))
        if self.prof: add(I("s_memtime", self.s_stamp[3])); add(s_waitcnt(lgkm=0))
        for qb in range(2): add(self.epilogue(qb))
        if self.prof:
            X = self.vX
            add(s_waitcnt(vm=0)); add(I("s_memtime", self.s_stamp[4])); add(s_waitcnt(lgkm=0))
            add(I("s_cmp_eq_u64", self.s_prof, 0, ndef=0)); add(I("s_cbranch_scc1", "L_exit", ndef=0))
            # 8 dwords per wave: [t1-t0, t2-t1, t3-t2, t4-t3, n0, n1, nT, j] at prof + (wg * 4 + w) * 32
            for i in range(4):
                add(I("s_sub_u32", t[i], self.s_stamp[i + 1][0], self.s_stamp[i][0]))
            vals = [t[0], t[1], t[2], t[3], self.s_n0, self.s_n1, self.s_nT, self.s_j]
            for i, sv in enumerate(vals): add(I("v_mov_b32", X[i], sv))
            add(I("s_lshl_b32", t[4], self.s_wg, 2)); add(I("s_add_u32", t[4], t[4], self.s_w)); add(I("s_lshl_b32", t[4], t[4], 5))
            add(I("v_mov_b32", X[8], t[4]))
            add(I("v_cmp_eq_u32", self.s_m64, 0, self.vLANE))
            add(I("s_and_saveexec_b64", self.s_exec, self.s_m64, xdef=[EXEC]))
            add(I("global_store_dwordx4", X[8], X[0:4], self.s_prof, ndef=0))
            add(I("global_store_dwordx4", X[8], X[4:8], self.s_prof, ndef=0, mods="offset:16"))
            add(I("s_mov_b64", EXEC, self.s_exec))
            # iteration stamps behind the summary records: base = total_wgs * 4 * 32 bytes
            add(I("s_lshl_b32", t[5], self.s_total, 7)); add(I("s_lshl_b32", t[6], t[4], 5))      # (wg*4+w)*32 -> *1024
            add(I("s_add_u32", t[5], t[5], t[6]))
            add(I("v_lshlrev_b32", X[9], 2, self.vLANE)); add(I("v_add_u32", X[9], t[5], X[9]))
            for k in range(4):
                add(I("global_store_dword", X[9], self.vPROF[k], self.s_prof, ndef=0, mods=f"offset:{256 * k}"))
        P.label("L_exit")
        add(s_waitcnt(vm=0))
        add(I("s_endpgm", ndef=0))
        add(self.ool)
        return P

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
                I("s_lshl_b32", t[0], self.s_w, 13), I("s_add_u32", t[0], t[0], STAGING), I("v_add_u32", X[4], t[0], X[4])]
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
        # rows back out: instruction `it` covers rows 4 it + (lane >> 4), 16 B per lane
        out += [I("v_lshrrev_b32", X[0], 4, self.vLANE), I("v_and_b32", X[3], 15, self.vLANE),
                I("v_xor_b32", X[5], X[3], X[0]), I("v_lshlrev_b32", X[5], 4, X[5]), I("v_lshl_add_u32", X[5], X[0], 8, X[5]),
                I("v_add_u32", X[5], t[0], X[5]),
                # global: (rb + 4 it + (lane >> 4)) * H*256 + h*256 + (lane & 15) * 16
                I("v_mul_lo_u32", X[2], X[0], self.s_outrow), I("v_lshl_add_u32", X[2], X[3], 4, X[2]),
                I("s_mul_i32", t[1], rb, self.s_outrow), I("s_lshl_b32", t[2], self.s_h, 8), I("s_add_u32", t[1], t[1], t[2]),
                I("s_lshl_b32", t[3], self.s_outrow, 2)]
        for it in range(8):
            out += [I("v_xor_b32", X[4], (it & 3) << 6, X[5]),
                    I("ds_read_b128", X[8:12], X[4], mods=f"offset:{1024 * it}"),
                    s_waitcnt(lgkm=0),
                    I("buffer_store_dwordx4", X[8:12], X[2], self.s_ro, t[1], ndef=0, mods="offen"),
                    s_nop(1),
                    I("s_add_u32", t[1], t[1], t[3])]
        out += [s_waitcnt(vm=0), Label(f"L_epi_skip_{st}")]
        return fix_hazards(out, f"epilogue{qb}", self.report)


def generate(debug_serial=False, abl=(), prof=False, name=KNAME):
    g = Gen(debug_serial=debug_serial, abl=abl, prof=prof)
    prog = g.build()
    items = fix_hazards(prog.items, "whole", g.report)
    body = render(items)
    hdr = ("// GENERATED by csrc/asm/attn_fwd_d128.py — do not edit.  Register map:\n" + g.S.listing() + "\n" + g.V.listing() + "\n"
           + g.A.listing() + "\n// hazard pads inserted: " + ", ".join(f"{t}={n}" for t, n in g.report))
    n_mfma = sum(1 for x in items if isinstance(x, Ins) and x.is_mfma)
    args = ARGS + ([(8, 64, "global_buffer")] if prof else [])
    text = kernel_text(name, body, lds_bytes=LDS_BYTES, kernarg_bytes=KERNARG_BYTES + (8 if prof else 0), n_arch_vgpr=g.V.next,
                       n_acc_vgpr=256, n_sgpr=g.S.next + 2, args_meta=args, header_comment=hdr)
    return text, g, n_mfma


# development variants (make DEV=1): cycle stamps and timing-only ablations, selected by the switch attn_fwd_asm = 2 + index
DEV_VARIANTS = [("prof", dict(prof=True)), ("nosm", dict(prof=True, abl=("nosm",))), ("nodma", dict(prof=True, abl=("nodma",))),
                ("nolds", dict(prof=True, abl=("nolds",))), ("nomfma", dict(prof=True, abl=("nomfma",))),
                ("mfmaonly", dict(prof=True, abl=("nosm", "nodma", "nolds"))), ("noqk", dict(prof=True, abl=("noqk",))),
                ("nopv", dict(prof=True, abl=("nopv",))), ("nodmak", dict(prof=True, abl=("nodmak",))),
                ("nodmav", dict(prof=True, abl=("nodmav",)))]


def unique_labels(text, suffix):
    import re
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
