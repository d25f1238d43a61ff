#!/usr/bin/env python3
"""CLIP attention forward (CLIPAttention of the frozen ViT-L/14 tower: non-causal, dh = 64, S = 257 tokens, fp16) for gfx950 —
hand-allocated registers, one wave per SIMD, persistent.  Replaces attn_fwd_resident_kernel<64, true, 257> of attention.hip
(llava/model/multimodal_encoder/clip_encoder.py:39-51 -> transformers CLIPAttention; the reference runs the tower in fp16,
handsonvlm/train/train.py:316): same operands (fused qkv [frames * 257, 3 * H * 64] fp16), same results to rounding.

Why a second structure and not the decoder kernel's: an item (frame, head) is only 257 x 257 scores, so K and V of the item stay
RESIDENT in LDS (72 KiB, two buffers: the next item's K / V arrive by LDS-DMA while this one computes) and the nine 32-key steps
of an item are fully unrolled — every LDS offset is an immediate, no loop control, the tail mask exists only in the ninth step.
At dh = 64 a 32 x 32 score block costs 8 MFMAs (256 cycles) against ~61 VALU instructions: the kernel is VALU-issue-bound by
construction (profiles/r03d: the C++ kernel's two waves per SIMD do not overlap each other's VALU and MFMA phases), so everything is
arranged for ONE instruction stream per SIMD that never stalls on the matrix pipe:
  * workgroup = 4 waves, wave w owns query rows [32 w, 32 w + 32) and [128 + 32 w, 160 + 32 w): eight full blocks, balanced;
  * three-stage pipeline inside an item:  QK^T of step t+2 | softmax of step t+1 | P V of step t  (as attn_fwd_d128.py);
  * the 257th query (the class token's row) is computed by wave 3 with SWAPPED MFMA operands (lanes = keys: one score register
    per step, 9 exponentials per lane instead of 144), reusing the K / V fragments already in registers; its probabilities reach
    the k-slot order of the V^T fragments through a 64-byte LDS strip.  The other three waves wait ~4100 cycles per item for it.
    A BALANCED form exists as the dev variant `split` (solo_split=True: every wave computes the scores of two or three key steps
    in the prologue, one barrier, the 257-key softmax per wave, P V split into (key half, head-dim half) = wave, partial sums
    merged through LDS after the next item's barrier — one code path for the four waves, half the code).  Measured
    (profiles/r05m_clip_attn_split.txt, r05z_attn_variant_energy.txt): every wave 14.2k cycles of body instead of 11.9k / 15.2k, the same
    17.3k cycles per item; on a settled chip 927-929 vs 924-926 us and +0.6 % joules per call (5 % slower in 10-launch timing loops,
    which do not settle the power manager) — it issues 84 instead of 72 class-row MFMAs and four softmaxes instead of one, and that
    costs what the balance gains.  Not adopted;
  * deferred running maximum (raised only when exceeded by 2^8) with the O rescale out of line, as in the decoder kernel;
  * outputs leave as 16-byte stores after a v_permlane32_swap exchange of the accumulator halves.
"""
import re
import sys
from asmgen import (Reg, RegFile, Program, I, Ins, Label, Raw, Filler, VCC, M0, EXEC, s_nop, s_waitcnt, interleave,
                    fix_hazards, add_lgkm_waits, render, kernel_text)

KNAME = "hvlm_fa_clip_d64_s257"
THR_BITS = 0x41000000
NEG_INF = 0xff800000
NKB = 9                         # 32-key steps per item (257 keys, the ninth holds one)

KBYTES = NKB * 4096             # K image: row major [288][64] fp16 like V (whole 128-byte rows per DMA lane group: the [ks][hi][key] image of the
                                # decoder kernel fetches 32-byte granules of 32 rows per instruction — measured slower here, the kernel is HBM-bound)
VBYTES = NKB * 32 * 128         # V image: row major [288][64] fp16, 16-B chunks XOR-swizzled by row (attention.hip swz_bits<64>)
BUF = KBYTES + VBYTES           # 73728
STRIP = 2 * BUF                 # 2 x 64 B: wave 3's probabilities of the class-token row
SSTRIP = STRIP + 256            # class-token row, `split` variant: raw scores [12 key steps][32] f32 (steps 9..11: the idle third rounds)
PSTRIP = SSTRIP + 1536          # its normalised probabilities [288] fp16, key order
OSTRIP = PSTRIP + 576 + 64      # its partial outputs [item parity][wave][hi][16] f32
LDS_BYTES = OSTRIP + 1024

ARGS = [(8, 0, "global_buffer"), (8, 8, "global_buffer"), (4, 16, "by_value"), (4, 20, "by_value"), (4, 24, "by_value"), (4, 28, "by_value"),
        (4, 32, "by_value"), (4, 36, "by_value"), (4, 40, "by_value"), (4, 44, "by_value")]
KERNARG_BYTES = 48


def mfma_f16(d, a, b, c):
    assert d.n == 16 and a.n == 4 and b.n == 4
    return I("v_mfma_f32_32x32x16_f16", d, a, b, c)


class Gen:
    def __init__(self, abl=(), prof=False, dma_steps=5, solo_split=False, dot_sum=False, lazy_max=True):
        self.abl, self.prof, self.dma_steps, self.split, self.dot_sum, self.lazy_max = set(abl), prof, dma_steps, solo_split, dot_sum, lazy_max
        self.report, self.site, self.ool = [], 0, []
        S = self.S = RegFile("s", 0, 102)
        V = self.V = RegFile("v", 0, 256)
        A = self.A = RegFile("a", 0, 256)
        self.s_karg = S.alloc("kernarg", 2); self.s_wg = S.alloc("workgroup id", 1); self.s_w = S.alloc("wave", 1)
        self.s_rq = S.alloc("rsrc qkv", 4, 4); self.s_ro = S.alloc("rsrc out", 4, 4)
        self.s_args = S.alloc("args: qkv, out, n_items, H, ld, scale, G, magic_H, qkv_bytes, out_bytes", 12, 4)
        self.s_qkv, self.s_out = self.s_args.sub(0, 2), self.s_args.sub(2, 2)
        (self.s_nitems, self.s_H, self.s_ld, self.s_scale, self.s_G, self.s_magh, self.s_qbytes, self.s_obytes) = [self.s_args[4 + i] for i in range(8)]
        self.s_t = [S.alloc(f"tmp{i}", 1) for i in range(8)]
        self.s_item = S.alloc("item", 1); self.s_next = S.alloc("next item", 1)
        self.s_ld2 = S.alloc("ld * 2", 1); self.s_outrow = S.alloc("H * 128", 1)
        self.s_qbase = S.alloc("this item: byte offset of its first row + h * 128 in qkv", 1)
        self.s_obase = S.alloc("this item: byte offset in out", 1)
        self.s_nbase = S.alloc("next item: qkv byte offset", 1)
        self.s_kcol = S.alloc("H * 128 (K column offset)", 1)
        self.s_buf = S.alloc("LDS buffer of this item (0 / BUF)", 1); self.s_nbuf = S.alloc("LDS buffer of the next item", 1)
        self.s_delta = S.alloc("address step to the other buffer", 1)
        self.s_pend = S.alloc("pending O rescale bits", 1)
        self.s_so = S.alloc("soffset tmp", 1)
        self.s_vtail = S.alloc("V soffset of row 256", 1); self.s_ktail = S.alloc("K soffset of row 256", 1)
        self.s_exec = S.alloc("saved exec", 2, 2)
        self.s_ones = S.alloc("fp16 (1.0, 1.0): the row sum as a dot product of the ROUNDED probability pairs (dot_sum)", 1)
        if solo_split:
            self.s_oprev = S.alloc("previous item: byte offset in out", 1); self.s_havep = S.alloc("a previous item's class row is pending", 1)
            self.s_opar = S.alloc("LDS parity offset of this item's partial class-row outputs (0 / 512)", 1)
        if prof:
            self.s_prof = S.alloc("prof", 2, 2); self.s_tm = self.s_karg; self.s_t0 = S.alloc("previous stamp", 1)
        # AccVGPRs
        self.aO = [A.alloc(f"O qb{q}", 32) for q in range(2)]
        self.aQ = [A.alloc(f"Q qb{q}", 16) for q in range(2)]
        self.aK = A.alloc("K fragments (4 ks x 4)", 16)
        self.aV = A.alloc("V fragments (2 sp x 2 db x 4)", 16)
        self.aOS = A.alloc("O of the class-token row", 16 if solo_split else 32)
        self.aQS = A.alloc("Q of the class-token row", 16)
        if solo_split:
            self.aKS = A.alloc("K fragments of the class row's score rounds", 16)
            self.aVS = [A.alloc(f"V fragment (sp_w, db_w) of the class row's P V, parity {i}", 4) for i in range(2)]
        self.vQN = A.alloc("next item's Q (2 qb x 16) + solo (16): prefetched into AccVGPRs", 48, 4)
        # arch VGPRs
        self.v_tid = V.alloc("tid", 1)
        self.vS = [[V.alloc(f"S qb{q} p{p}", 16, 4) for p in range(2)] for q in range(2)]
        self.vP = [[V.alloc(f"P qb{q} p{p}", 8, 4) for p in range(2)] for q in range(2)]
        self.vSS = [V.alloc(f"S solo p{p}", 16, 4) for p in range(1 if solo_split else 2)]
        self.vPS = [V.alloc(f"P solo p{p}", 4 if solo_split else 8, 4) for p in range(2)]
        if solo_split:
            self.vKSADDR = [V.alloc(f"class row: K fragment address of key step w, ks{ks}", 1) for ks in range(4)]
            self.vVSADDR = [V.alloc(f"class row: V tr-read address (sp_w, db_w) hf{hf}", 1) for hf in range(2)]
            self.vSWR = V.alloc("class row: score write address", 1); self.vSRD = V.alloc("class row: score read address", 1)
            self.vPWR = V.alloc("class row: P write address", 1); self.vPRD = V.alloc("class row: P fragment read address", 1)
            self.vOWR = V.alloc("class row: partial O write / read address", 1); self.vOSOFF = V.alloc("class row: out byte offset of the lane", 1)
            self.vSC = V.alloc("class row: score copies of the three rounds", 3)
        self.vM = [V.alloc(f"m {n}", 1) for n in ("qb0", "qb1", "solo")]
        self.vNEGM = [V.alloc(f"-m {n}", 1) for n in ("qb0", "qb1", "solo")]
        self.vMT = [V.alloc(f"m+THR {n}", 1) for n in ("qb0", "qb1", "solo")]
        if lazy_max:
            self.vMTR = [V.alloc(f"(m+THR-0.5)/scale {n}: the trigger threshold in raw-score units", 1) for n in ("qb0", "qb1")]
            self.vRSC = V.alloc("1 / scale_log2", 1)
        self.vL = [V.alloc(f"l {n}", 1) for n in ("qb0", "qb1", "solo")]
        self.vALPHA = [V.alloc(f"pending alpha {n}", 1) for n in ("qb0", "qb1", "solo")]
        self.vT = [V.alloc(f"softmax temps {n}", 6) for n in ("qb0", "qb1", "solo")]
        self.vACC = [V.alloc(f"partial sums qb{q}", 4) for q in range(2)]
        self.vNINF = V.alloc("-inf", 1)
        self.vKADDR = [V.alloc(f"K fragment LDS address ks{ks}", 1) for ks in range(4)]
        self.vVADDR = [[V.alloc(f"V tr-read address db{db} hf{hf}", 1) for hf in range(2)] for db in range(2)]
        self.vSTRIP = V.alloc("strip write / read addresses", 2)
        self.vKOFF = V.alloc("K DMA voffset", 1); self.vVOFF = V.alloc("V DMA voffset", 1)
        self.vKOFFT = V.alloc("K DMA voffset, rows clamped to row 256 (ninth step)", 1); self.vVOFFT = V.alloc("V DMA voffset, rows clamped to row 256", 1)
        self.vLANE = V.alloc("lane", 1); self.vL31 = V.alloc("lane & 31", 1); self.vHI = V.alloc("lane >> 5", 1)
        self.vQOFF = [V.alloc(f"Q row byte offset qb{q}", 1) for q in range(2)] + [V.alloc("Q row byte offset solo", 1)]
        self.vOOFF = [V.alloc(f"O row byte offset qb{q}", 1) for q in range(2)]
        self.vX = V.alloc("scratch", 12, 4)
        self.vPACC = V.alloc("prof sums", 1)

    def new_site(self):
        self.site += 1
        return self.site

    # ---- fragments / MFMAs -------------------------------------------------------------------------------------------
    def k_read(self, t, ks):
        return I("ds_read_b128", self.aK.sub(4 * ks, 4), self.vKADDR[ks], mods=f"offset:{t * 4096}")

    def v_reads(self, t, sp, db):
        off = (32 * t + 16 * sp) * 128
        base = self.aV.sub(4 * (2 * sp + db), 4)
        return [I("ds_read_b64_tr_b16", base.sub(2 * hf, 2), self.vVADDR[db][hf], mods=f"offset:{off}") for hf in range(2)]

    def qk(self, qb, p, ks):
        d = self.vS[qb][p]
        return mfma_f16(d, self.aK.sub(4 * ks, 4), self.aQ[qb].sub(4 * ks, 4), 0 if ks == 0 else d)

    def pv(self, qb, p, sp, db):
        o = self.aO[qb].sub(16 * db, 16)
        return mfma_f16(o, self.aV.sub(4 * (2 * sp + db), 4), self.vP[qb][p].sub(4 * sp, 4), o)

    def qk_solo(self, p, ks):                       # swapped: A = the query (all 32 rows the same row), B = K -> lanes are KEYS
        d = self.vSS[p]
        return mfma_f16(d, self.aQS.sub(4 * ks, 4), self.aK.sub(4 * ks, 4), 0 if ks == 0 else d)

    def pv_solo(self, p, sp, db):
        o = self.aOS.sub(16 * db, 16)
        return mfma_f16(o, self.aV.sub(4 * (2 * sp + db), 4), self.vPS[p].sub(4 * sp, 4), o)

    # ---- DMA of the next item's K / V: 18 pieces per wave, (m0, soffset, load) groups ------------------------------------------
    def dma_groups(self):
        """pieces of this wave: K ids w, w+4, ... (9) and V ids w, w+4, ... (9).  K piece id = (t, ks) = (id >> 2, id & 3) at
        LDS nbuf + id * 1024, source row 32 t, column 32 ks bytes; V piece pi at nbuf + KBYTES + pi * 1024, source row 8 pi."""
        g = []
        t = self.s_t
        # (rows past 256 belong to the next frame — or to nothing, behind the last one: the ninth pieces read row 256 for every lane.
        #  The masked keys' K is never used; their V must be FINITE, it is multiplied by P = 0.)
        for k in range(9):
            # K: piece pi = w + 4 k -> rows 8 w + 32 k (same image and source pattern as V)
            g.append([I("s_add_u32", M0, self.s_t[6], 4096 * k),                                   # t6 = nbuf + w * 1024
                      (I("s_add_u32", self.s_so, self.s_t[4], 0) if k == 0 else I("s_add_u32", self.s_so, self.s_so, self.s_t[5])) if k < 8
                      else I("s_mov_b32", self.s_so, self.s_ktail),                                # t4 = K base, t5 = 32 rows
                      I("buffer_load_dwordx4", self.vVOFF if k < 8 else self.vVOFFT, self.s_rq, self.s_so, ndef=0, mods="offen lds", xuse=[M0])])
        for k in range(9):
            # V: pi = w + 4 k -> rows 8 w + 32 k
            g.append([I("s_add_u32", M0, self.s_t[7], 4096 * k),                                   # t7 = nbuf + KBYTES + w * 1024
                      (I("s_add_u32", self.s_t[3], self.s_t[2], 0) if k == 0 else I("s_add_u32", self.s_t[3], self.s_t[3], self.s_t[5])) if k < 8
                      else I("s_mov_b32", self.s_t[3], self.s_vtail),                              # t2 = V base
                      I("buffer_load_dwordx4", self.vVOFF if k < 8 else self.vVOFFT, self.s_rq, self.s_t[3], ndef=0, mods="offen lds", xuse=[M0])])
        return g

    def dma_setup(self, item_base):
        """scalar state of dma_groups for the item whose qkv byte offset (first row, + h * 128) is item_base."""
        t = self.s_t
        return [I("s_lshl_b32", t[0], self.s_w, 10),
                I("s_add_u32", t[6], self.s_nbuf, t[0]), I("s_add_u32", t[7], t[6], KBYTES),
                I("s_lshl_b32", t[5], self.s_ld2, 5),                                              # 32 rows
                I("s_lshl_b32", t[1], self.s_ld2, 3), I("s_mul_i32", t[1], t[1], self.s_w),        # rows 8 w
                I("s_add_u32", t[4], item_base, self.s_kcol),
                I("s_lshl_b32", self.s_ktail, self.s_ld2, 8), I("s_add_u32", self.s_ktail, self.s_ktail, t[4]),   # row 256 of K
                I("s_add_u32", t[4], t[4], t[1]),
                I("s_add_u32", t[2], item_base, self.s_kcol), I("s_add_u32", t[2], t[2], self.s_kcol),
                I("s_lshl_b32", self.s_vtail, self.s_ld2, 8), I("s_add_u32", self.s_vtail, self.s_vtail, t[2]),   # row 256 of V
                I("s_add_u32", t[2], t[2], t[1])]

    # ---- softmax of one 32 x 32 block (S[qb][p] -> P[qb][p]) ---------------------------------------------------------------
    def sm_stream(self, qb, p, init, tail_mask):
        x = self.vS[qb][p]; P = self.vP[qb][p]; t = self.vT[qb]; acc = self.vACC[qb]
        m, negm, mt, l = self.vM[qb], self.vNEGM[qb], self.vMT[qb], self.vL[qb]
        sc = self.s_scale
        F = []
        def f(*ins, **kw): F.append(Filler(list(ins), **kw))
        if tail_mask:                                  # step 8: only key 256 = (i = 0, hi = 0) exists
            f(I("v_cmp_eq_u32", VCC, 0, self.vHI), I("v_cndmask_b32", x[0], self.vNINF, x[0], VCC))
            for i in range(1, 16): f(I("v_mov_b32", x[i], self.vNINF))
        f(I("v_max3_f32", t[0], x[0], x[1], x[2])); f(I("v_max3_f32", t[1], x[3], x[4], x[5]))
        f(I("v_max3_f32", t[2], x[6], x[7], x[8])); f(I("v_max3_f32", t[3], x[9], x[10], x[11]))
        f(I("v_max3_f32", t[4], x[12], x[13], x[14]))
        f(I("v_max3_f32", t[0], t[0], t[1], t[2])); f(I("v_max3_f32", t[3], t[3], t[4], x[15]))
        f(I("v_max_f32", t[0], t[0], t[3]))
        lazy = self.lazy_max and not init
        full = [I("v_mov_b32", t[1], t[0]), I("v_permlane32_swap_b32", t[1], t[0], ndef=2), I("v_max_f32", t[0], t[0], t[1]), I("v_mul_f32", t[0], sc, t[0])]
        if not lazy:
            for ins in full: f(ins)
        if init:
            f(I("v_mov_b32", m, t[0]), I("v_xor_b32", negm, 0x80000000, t[0]), I("v_add_f32", mt, THR_BITS, t[0]))
            if self.lazy_max: f(I("v_add_f32", t[1], -0.5, mt), I("v_mul_f32", self.vMTR[qb], self.vRSC, t[1]))
        else:
            st = self.new_site()
            # lazy_max (default; as attn_fwd_d128.py's option): the trigger tests the lane's PARTIAL raw maximum against (m + THR - 0.5) / scale; the slow path
            # repeats the exact per-lane test on the row's full, scaled maximum — bit-identical results, 4 VALU fewer per block-step: 910 vs 920 us and -1 % joules
            # on a settled chip (profiles/r06c_attn_variant_energy.txt; dev variant `eagermax` = the exact test on the fast path)
            f(I("v_cmp_gt_f32", VCC, t[0], self.vMTR[qb] if lazy else mt), I("s_cbranch_vccnz", f"L_slow_{st}", ndef=0, xuse=[VCC]), Label(f"L_slowret_{st}"))
            a = t[2]
            self.ool += [Label(f"L_slow_{st}")] + ((full + [I("v_cmp_gt_f32", VCC, t[0], mt)]) if lazy else []) + [
                         I("v_cndmask_b32", t[1], m, t[0], VCC), I("v_sub_f32", a, m, t[1]), I("v_exp_f32", a, a),
                         I("v_mov_b32", m, t[1]), I("v_xor_b32", negm, 0x80000000, t[1]), I("v_add_f32", mt, THR_BITS, t[1]),
                         I("v_mul_f32", l, l, a), I("v_mul_f32", self.vALPHA[qb], self.vALPHA[qb], a)] + (
                         [I("v_add_f32", t[1], -0.5, mt), I("v_mul_f32", self.vMTR[qb], self.vRSC, t[1])] if lazy else []) + [
                         I("s_bitset1_b32", self.s_pend, qb, rw=True), I("s_branch", f"L_slowret_{st}", ndef=0)]
        consumers = []
        if self.dot_sum:
            # row sum over the fp16-ROUNDED pairs (what P V multiplies), one v_dot2_f32_f16 per pair instead of two v_add_f32 on the fp32 values
            tail = [I("v_add_f32", acc[0], acc[0], acc[1]), I("v_add_f32", l, l, acc[0])]
        else:
            for k, (a_, b_) in enumerate([(0, 1), (2, 3), (4, 5), (6, 7)]): consumers.append((b_, I("v_add_f32", acc[k], x[a_], x[b_])))
            for i in range(8, 16): consumers.append((i, I("v_add_f32", acc[i & 3], acc[i & 3], x[i])))
            tail = [I("v_add_f32", acc[0], acc[0], acc[1]), I("v_add_f32", acc[2], acc[2], acc[3]), I("v_add_f32", acc[0], acc[0], acc[2]),
                    I("v_add_f32", l, l, acc[0])]
        order = sorted(range(8), key=lambda j: 8 * (j >> 2) + 2 * (j & 3) + 1)
        for n_, j in enumerate(order):
            sp, k = j >> 2, j & 3
            need = 8 * sp + 2 * k + 1
            consumers.append((need, I("v_cvt_pk_f16_f32", P[j], x[8 * sp + 2 * k], x[8 * sp + 2 * k + 1])))
            if self.dot_sum:
                a_ = acc[n_ & 1]
                consumers.append((need + 0.5, I("v_dot2_f32_f16", a_, P[j], self.s_ones, 0 if n_ < 2 else a_)))
        consumers.sort(key=lambda c: c[0])
        for i in range(16): f(I("v_fma_f32", x[i], x[i], sc, negm))
        ci = 0
        for i in range(16):
            f(I("v_exp_f32", x[i], x[i]))
            while ci < len(consumers) and consumers[ci][0] <= i - 2:
                f(consumers[ci][1]); ci += 1
                if ci % 2 == 0: break
        while ci < len(consumers): f(consumers[ci][1]); ci += 1
        for ins in tail: f(ins)
        return F

    # ---- softmax of the class-token row on one step (wave 3): lanes are keys, one score register -------------------------------
    def sm_solo(self, p, init, tail_mask):
        s = self.vSS[p][0]; t = self.vT[2]
        m, negm, mt, l = self.vM[2], self.vNEGM[2], self.vMT[2], self.vL[2]
        sc = self.s_scale
        F = []
        def f(*ins, **kw): F.append(Filler(list(ins), **kw))
        if tail_mask:                                  # only key 256 = lane 0 of either half
            f(I("v_cmp_eq_u32", VCC, 0, self.vL31), I("v_cndmask_b32", s, self.vNINF, s, VCC))
        # wave-wide maximum (both half-waves hold every key once)
        f(I("v_mov_b32", t[0], s))
        for ctrl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
            f(s_nop(1), I("v_max_f32_dpp", t[0], t[0], t[0], mods=f"{ctrl} row_mask:0xf bank_mask:0xf", rw=True))
        f(s_nop(0), I("v_readlane_b32", self.s_t[0], t[0], 0), I("v_readlane_b32", self.s_t[1], t[0], 16))
        f(I("v_mov_b32", t[1], self.s_t[1]), I("v_max_f32", t[0], self.s_t[0], t[1]))
        f(I("v_mul_f32", t[0], sc, t[0]))
        if init:
            f(I("v_mov_b32", m, t[0]), I("v_xor_b32", negm, 0x80000000, t[0]), I("v_add_f32", mt, THR_BITS, t[0]))
        else:
            st = self.new_site()
            f(I("v_cmp_gt_f32", VCC, t[0], mt), I("s_cbranch_vccnz", f"L_slow_{st}", ndef=0, xuse=[VCC]), Label(f"L_slowret_{st}"))
            a = t[2]
            self.ool += [Label(f"L_slow_{st}"), I("v_cndmask_b32", t[1], m, t[0], VCC), I("v_sub_f32", a, m, t[1]), I("v_exp_f32", a, a),
                         I("v_mov_b32", m, t[1]), I("v_xor_b32", negm, 0x80000000, t[1]), I("v_add_f32", mt, THR_BITS, t[1]),
                         I("v_mul_f32", l, l, a), I("v_mul_f32", self.vALPHA[2], self.vALPHA[2], a),
                         I("s_bitset1_b32", self.s_pend, 2, rw=True), I("s_branch", f"L_slowret_{st}", ndef=0)]
        f(I("v_fma_f32", t[3], s, sc, negm)); f(I("v_exp_f32", t[3], t[3]))
        f(s_nop(0), I("v_add_f32", l, l, t[3]), I("v_cvt_f16_f32", t[4], t[3]))
        # probabilities -> strip (key order) -> the k-slot order of the V^T fragments: keys b + {0..3}, b + 8 + {0..3}, b = 16 sp + 4 hi
        so = 64 * p
        f(I("ds_write_b16", self.vSTRIP[0], t[4], ndef=0, mods=f"offset:{so}"))
        for sp in range(2):
            d = self.vPS[p].sub(4 * sp, 4)
            f(I("ds_read_b64", d.sub(0, 2), self.vSTRIP[1], mods=f"offset:{so + 32 * sp}"),
              I("ds_read_b64", d.sub(2, 2), self.vSTRIP[1], mods=f"offset:{so + 32 * sp + 16}"))
        return F

    # ---- class-token row, shared by the four waves (solo_split) --------------------------------------------------------------
    def solo_scores(self):
        """Prologue: three rounds (key steps w, w + 4, w + 8: the third is real on wave 0 only, the others read past the K image —
        finite garbage into strip slots nobody reads) of 4 K fragment reads + 4 swapped MFMAs; raw score of key l31 -> strip."""
        out = []
        d = self.vSS[0]
        for i in range(3):
            for ks in range(4):
                out.append(I("ds_read_b128", self.aKS.sub(4 * ks, 4), self.vKSADDR[ks], mods=f"offset:{i * 16384}"))
            for ks in range(4):
                out.append(mfma_f16(d, self.aQS.sub(4 * ks, 4), self.aKS.sub(4 * ks, 4), 0 if ks == 0 else d))
            out.append(I("v_mov_b32", self.vSC[i], d[0]))
            out.append(I("ds_write_b32", self.vSWR, self.vSC[i], ndef=0, mods=f"offset:{i * 512}"))
        return out

    def solo_softmax(self):
        """Step 0: softmax of the class row over all 257 keys, by every wave for itself (lane l31 of either half: keys 32 tau + l31).
        Returns (the nine score reads, VALU stream).  Probabilities are NORMALISED before the fp16 rounding: P V needs no division."""
        sc = [self.vSS[0][i] for i in range(9)]
        t = self.vT[2]
        scal = self.s_scale
        # (the reads head the item body as plain instructions, NOT as `early` fillers: the interleaver orders fillers of different
        #  kinds by gap capacity only, and the first version's VALU stream reached the scores before their reads were issued)
        early = [I("ds_read_b32", sc[tau], self.vSRD, mods=f"offset:{tau * 128}") for tau in range(9)]
        F = []
        def f(*ins, **kw): F.append(Filler(list(ins), **kw))
        f(I("v_cmp_eq_u32", VCC, 0, self.vL31), I("v_cndmask_b32", sc[8], self.vNINF, sc[8], VCC))     # key step 8: only key 256 exists
        f(I("v_max3_f32", t[0], sc[0], sc[1], sc[2])); f(I("v_max3_f32", t[1], sc[3], sc[4], sc[5])); f(I("v_max3_f32", t[2], sc[6], sc[7], sc[8]))
        f(I("v_max3_f32", t[0], t[0], t[1], t[2]))
        for ctrl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
            f(s_nop(1), I("v_max_f32_dpp", t[0], t[0], t[0], mods=f"{ctrl} row_mask:0xf bank_mask:0xf", rw=True))
        f(s_nop(0), I("v_readlane_b32", self.s_t[0], t[0], 0), I("v_readlane_b32", self.s_t[1], t[0], 16))
        f(I("v_mov_b32", t[1], self.s_t[1]), I("v_max_f32", t[0], self.s_t[0], t[1]))
        f(I("v_mul_f32", t[0], scal, t[0])); f(I("v_xor_b32", t[1], 0x80000000, t[0]))                 # t1 = -m * scale
        for tau in range(9): f(I("v_fma_f32", sc[tau], sc[tau], scal, t[1]))
        for tau in range(9): f(I("v_exp_f32", sc[tau], sc[tau]))
        f(s_nop(0), I("v_add_f32", t[2], sc[0], sc[1])); f(I("v_add_f32", t[3], sc[2], sc[3])); f(I("v_add_f32", t[2], t[2], sc[4])); f(I("v_add_f32", t[3], t[3], sc[5]))
        f(I("v_add_f32", t[2], t[2], sc[6])); f(I("v_add_f32", t[3], t[3], sc[7])); f(I("v_add_f32", t[2], t[2], sc[8])); f(I("v_add_f32", t[2], t[2], t[3]))
        for ctrl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
            f(s_nop(1), I("v_add_f32_dpp", t[2], t[2], t[2], mods=f"{ctrl} row_mask:0xf bank_mask:0xf", rw=True))
        f(s_nop(0), I("v_readlane_b32", self.s_t[0], t[2], 0), I("v_readlane_b32", self.s_t[1], t[2], 16))
        f(I("v_mov_b32", t[3], self.s_t[1]), I("v_add_f32", t[2], self.s_t[0], t[3]), I("v_rcp_f32", t[3], t[2]))
        f(s_nop(0), I("v_mul_f32", sc[0], sc[0], t[3]))
        for tau in range(1, 9): f(I("v_mul_f32", sc[tau], sc[tau], t[3]))
        for tau in range(9):
            f(I("v_cvt_f16_f32", sc[tau], sc[tau]))
        for tau in range(9):
            f(I("ds_write_b16", self.vPWR, sc[tau], ndef=0, mods=f"offset:{tau * 64}"))
        return early, F

    def solo_pv(self, tau):
        """(LDS reads, MFMA): the class row's P V on key step tau, this wave's (key half sp_w, head-dim half db_w)."""
        i = tau & 1
        rd = [I("ds_read_b64_tr_b16", self.aVS[i].sub(2 * hf, 2), self.vVSADDR[hf], mods=f"offset:{tau * 4096}") for hf in range(2)]
        rd += [I("ds_read_b64", self.vPS[i].sub(0, 2), self.vPRD, mods=f"offset:{64 * tau}"),
               I("ds_read_b64", self.vPS[i].sub(2, 2), self.vPRD, mods=f"offset:{64 * tau + 16}")]
        return rd, mfma_f16(self.aOS, self.aVS[i], self.vPS[i], self.aOS)

    def solo_partial_write(self):
        """End of an item: this wave's partial class-row output (16 registers of lanes l31 == 0) -> LDS [parity][wave][hi][16]."""
        T = self.vS[0][0]
        out = [I("v_accvgpr_read_b32", T[k], self.aOS[k]) for k in range(16)]
        out += [I("v_add_u32", self.vX[0], self.s_opar, self.vOWR)]
        out += [I("v_cmp_eq_u32", VCC, 0, self.vL31), I("s_and_saveexec_b64", self.s_exec, VCC, xdef=[EXEC])]
        out += [I("ds_write_b128", self.vX[0], T.sub(4 * k, 4), ndef=0, mods=f"offset:{16 * k}") for k in range(4)]
        out += [I("s_mov_b64", EXEC, self.s_exec)]
        return fix_hazards(out, "solo_partial", self.report)

    def solo_merge(self, tag):
        """Behind a workgroup barrier: waves 0 / 1 (sp = 0, db = w) add their own and wave w + 2's partial of the PREVIOUS item
        (parity s_opar) and store head dims 32 db + 8 k + 4 hi + {0..3} of row 256: lanes l31 == 0, 8 bytes per k."""
        A_, B_ = self.vS[0][0], self.vS[0][1]
        X = self.vX
        t = self.s_t
        out = [I("s_cmp_eq_u32", self.s_havep, 0, ndef=0), I("s_cbranch_scc1", f"L_mrg_skip_{tag}", ndef=0),
               I("s_cmp_ge_u32", self.s_w, 2, ndef=0), I("s_cbranch_scc1", f"L_mrg_skip_{tag}", ndef=0)]
        body = [I("v_add_u32", X[8], self.s_opar, self.vOWR)]
        body += [I("v_cmp_eq_u32", VCC, 0, self.vL31), I("s_and_saveexec_b64", self.s_exec, VCC, xdef=[EXEC])]
        body += [I("ds_read_b128", A_.sub(4 * k, 4), X[8], mods=f"offset:{16 * k}") for k in range(4)]
        body += [I("ds_read_b128", B_.sub(4 * k, 4), X[8], mods=f"offset:{256 + 16 * k}") for k in range(4)]
        body += [I("s_lshl_b32", t[2], self.s_outrow, 8), I("s_add_u32", t[2], t[2], self.s_oprev)]            # row 256 of the previous item
        for k in range(4):
            body += [I("v_add_f32", A_[4 * k + j], A_[4 * k + j], B_[4 * k + j]) for j in range(4)]
            body += [I("v_cvt_pk_f16_f32", X[2 * (k & 1)], A_[4 * k], A_[4 * k + 1]), I("v_cvt_pk_f16_f32", X[2 * (k & 1) + 1], A_[4 * k + 2], A_[4 * k + 3]),
                     I("buffer_store_dwordx2", X[2 * (k & 1): 2 * (k & 1) + 2], self.vOSOFF, self.s_ro, t[2], ndef=0, mods=f"offen offset:{16 * k}"), s_nop(1)]
        body += [I("s_mov_b64", EXEC, self.s_exec)]
        body, _ = add_lgkm_waits(body)
        out += fix_hazards(body, f"solo_merge_{tag}", self.report)
        out += [Label(f"L_mrg_skip_{tag}")]
        return out

    def rescale_check(self, solo):
        st = self.new_site()
        blk = [Label(f"L_resc_{st}"), s_nop(15)]
        tmp = self.vX
        for qb in (0, 1) + ((2,) if solo else ()):
            acc = self.aOS if qb == 2 else self.aO[qb]
            blk += [I("s_bitcmp1_b32", self.s_pend, qb, ndef=0), I("s_cbranch_scc0", f"L_resc_{st}_skip{qb}", ndef=0)]
            for r0 in range(0, 32, 8):
                for k in range(8): blk.append(I("v_accvgpr_read_b32", tmp[k], acc[r0 + k]))
                for k in range(8): blk.append(I("v_mul_f32", tmp[k], tmp[k], self.vALPHA[qb]))
                for k in range(8): blk.append(I("v_accvgpr_write_b32", acc[r0 + k], tmp[k]))
            blk += [I("v_mov_b32", self.vALPHA[qb], 1.0), Label(f"L_resc_{st}_skip{qb}")]
        blk += [I("s_mov_b32", self.s_pend, 0), s_nop(1), I("s_branch", f"L_rescret_{st}", ndef=0)]
        self.ool += blk
        return [I("s_cmp_lg_u32", self.s_pend, 0, ndef=0), I("s_cbranch_scc1", f"L_resc_{st}", ndef=0), Label(f"L_rescret_{st}")]

    # ---- step t of an item: QK(t+2) | softmax(t+1) | PV(t) ------------------------------------------------------------------
    def step(self, t, solo, dma):
        p = t & 1
        has_qk, has_sm = t + 2 < NKB, t + 1 < NKB
        mf, k_after, v_after = [], {}, {}
        for j in range(4):                             # slice j: K fragment ks = j, V fragment (sp, db) = (j >> 1, j & 1)
            sp, db = j >> 1, j & 1
            for qb in range(2):
                if has_qk: mf.append(self.qk(qb, p, j))
                mf.append(self.pv(qb, p, sp, db))
            if solo:                                   # the class-token row uses the same fragments: right behind the regular ones
                if has_qk: mf.append(self.qk_solo(p, j))
                mf.append(self.pv_solo(p, sp, db))
            last_qk = max([i for i, m_ in enumerate(mf) if m_.ops[1].idx == self.aK.idx + 4 * j or (len(m_.ops) > 2 and hasattr(m_.ops[2], "idx") and m_.ops[2].kind == "a" and m_.ops[2].idx == self.aK.idx + 4 * j)], default=len(mf) - 1)
            k_after[j] = last_qk
            v_after[(sp, db)] = len(mf) - 1
        early = []
        if "nolds" not in self.abl:
            for ks in range(4):
                if t + 3 < NKB and has_qk: early.append(Filler([self.k_read(t + 3, ks)], after=k_after[ks], prio=0))
            for sp in range(2):
                for db in range(2):
                    if t + 1 < NKB: early.append(Filler(self.v_reads(t + 1, sp, db), after=v_after[(sp, db)], prio=0))
        if dma and "nodma" not in self.abl:
            gi = 0
            for grp in dma:
                early.append(Filler(grp, after=min(2 * gi + 1, len(mf) - 1), prio=0)); gi += 1
        stream = []
        n_ool = len(self.ool)
        if has_sm:
            a = self.sm_stream(0, 1 - p, init=False, tail_mask=(t + 1 == NKB - 1))
            b = self.sm_stream(1, 1 - p, init=False, tail_mask=(t + 1 == NKB - 1))
            c = self.sm_solo(1 - p, init=False, tail_mask=(t + 1 == NKB - 1)) if solo else []
            for i in range(max(len(a), len(b))):
                if i < len(a): stream.append(a[i])
                if i < len(b): stream.append(b[i])
                if i % 4 == 3 and c: stream.append(c.pop(0))
            stream += c
        if self.split:
            if t == 0 and "nosm" not in self.abl:
                _, c = self.solo_softmax()            # (its score reads: item_body puts them in front of step 0)
                merged, ci = [], 0                    # the class row's softmax rides between the two blocks' streams
                for i, fl in enumerate(stream):
                    merged.append(fl)
                    if i % 3 == 2 and ci < len(c): merged.append(c[ci]); ci += 1
                stream = merged + c[ci:]
            for tau in ([t - 1] if 1 <= t < NKB - 1 else [NKB - 2, NKB - 1] if t == NKB - 1 else []):
                rd, m_ = self.solo_pv(tau)
                if "nolds" not in self.abl: early.append(Filler(rd, after=0, before=len(mf) - 1, prio=0))
                mf.append(m_)
        if "nosm" in self.abl: stream, self.ool = [], self.ool[:n_ool]
        if "nomfma" in self.abl: mf = [s_nop(0) for _ in mf]
        return self.rescale_check(solo) + interleave(mf, early, stream, head_room=2)

    # ---- one whole item for a wave (solo: wave 3) ----------------------------------------------------------------------------
    def item_body(self, P, solo):
        add = P.add
        X = self.vX
        tag = "s" if solo else "r"
        # prologue: QK of steps 0 and 1, fragments for step 0's body, softmax of step 0
        pro = []
        for t in range(2):
            for ks in range(4): pro.append(self.k_read(t, ks))
            for ks in range(4):
                for qb in range(2): pro.append(self.qk(qb, t, ks))
                if solo: pro.append(self.qk_solo(t, ks))
        if self.split: pro += self.solo_scores()
        for ks in range(4): pro.append(self.k_read(2, ks))
        for sp in range(2):
            for db in range(2): pro += self.v_reads(0, sp, db)
        pro, _ = add_lgkm_waits(pro)
        pro = fix_hazards(pro, f"prologue_{tag}", self.report)
        add(pro)
        add(s_nop(15))
        sm0 = []
        for qb in range(2): sm0 += [x for fl in self.sm_stream(qb, 0, init=True, tail_mask=False) for x in fl.ins]
        if solo: sm0 += [x for fl in self.sm_solo(0, init=True, tail_mask=False) for x in fl.ins]
        sm0, _ = add_lgkm_waits(sm0)
        add(sm0)
        add(s_waitcnt(lgkm=0))
        if self.split: add(I("s_barrier", ndef=0))     # every wave's class-row scores are in the strip
        # The kernel moves 2.5 GB of q | k | v in and 0.8 GB out per call of the tower's shape (>= 540 us at the 6.3 TB/s a streaming
        # kernel reaches): the next item's 18 pieces are spread over the first `dma_steps` steps of this item.  Measured
        # (profiles/r05f_clip_attn_asm.txt): all in step 0: 976 us, 2 steps 933, 3 steps 923, 5 steps 879, 9 steps 910 — a burst stalls
        # the issuing wave (in-order issue behind a full memory queue), the last pieces of a full spread land too late.
        dma = self.dma_groups()
        per = (len(dma) + self.dma_steps - 1) // self.dma_steps
        body = list(self.solo_softmax()[0]) if (self.split and "nosm" not in self.abl) else []
        for t in range(NKB):
            body += self.step(t, solo, dma[per * t: per * (t + 1)])
        # lgkm entry state: everything the prologue issued has been waited for (lgkmcnt(0) above)
        body, _ = add_lgkm_waits(body)
        body = fix_hazards(body, f"item_{tag}", self.report)
        add(body)

    # ---- epilogue: O / l -> fp16 -> 16-byte stores ----------------------------------------------------------------------------
    def epilogue(self, solo):
        X = self.vX
        out = [s_nop(15)]
        for qb in range(2):
            l = self.vL[qb]
            out += [I("v_mov_b32", X[0], l), s_nop(1), I("v_permlane32_swap_b32", X[0], l, ndef=2), I("v_add_f32", l, l, X[0]), I("v_rcp_f32", X[1], l), s_nop(0)]
            for db in range(2):
                for gp in range(2):                    # groups g = 2 gp, 2 gp + 1 (4 head dims each) -> one 16-byte store
                    pk = []
                    for g in (2 * gp, 2 * gp + 1):
                        src = self.aO[qb].sub(16 * db + 4 * g, 4)
                        out += [I("v_accvgpr_read_b32", X[8 + k], src[k]) for k in range(4)]
                        out += [I("v_mul_f32", X[8 + k], X[8 + k], X[1]) for k in range(4)]
                        a, b = X[2 + 2 * (g & 1)], X[3 + 2 * (g & 1)]
                        out += [I("v_cvt_pk_f16_f32", a, X[8], X[9]), I("v_cvt_pk_f16_f32", b, X[10], X[11])]
                        pk.append((a, b))
                    (a0, a1), (b0, b1) = pk            # group k = (a0, a1), group k+1 = (b0, b1): swap halves (cdna guide T21)
                    out += [s_nop(1), I("v_permlane32_swap_b32", a0, b0, ndef=2), I("v_permlane32_swap_b32", a1, b1, ndef=2)]
                    # lanes 0-31 now hold head dims 8k..8k+7 of their row in (a0, a1, b0, b1)?  lower: [own k | upper's k]; upper: [lower's k+1 | own k+1]
                    # lower lanes: (a0, a1) = own group k, (b0, b1) = upper's group k  -> 16 bytes at dims 8k .. 8k+7:  a0 a1 b0 b1
                    # upper lanes: (a0, a1) = lower's group k+1, (b0, b1) = own group k+1 -> dims 8(k+1) ..:              a0 a1 b0 b1
                    assert (a0.idx, a1.idx, b0.idx, b1.idx) == (X[2].idx, X[3].idx, X[4].idx, X[5].idx)
                    d0 = 32 * db + 16 * gp             # first head dim of the lower lanes' 16 bytes; upper lanes: + 8 dims = + 16 bytes
                    out += [I("buffer_store_dwordx4", X[2:6], self.vOOFF[qb], self.s_ro, self.s_obase, ndef=0, mods=f"offen offset:{2 * d0}"), s_nop(1)]
        if solo:
            l = self.vL[2]; t = self.vT[2]
            # row sum over the keys = lanes 0..31 (the other half holds the same values)
            out += [I("v_mov_b32", t[0], l)]
            for ctrl in ("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
                out += [s_nop(1), I("v_add_f32_dpp", t[0], t[0], t[0], mods=f"{ctrl} row_mask:0xf bank_mask:0xf", rw=True)]
            out += [s_nop(0), I("v_readlane_b32", self.s_t[0], t[0], 0), I("v_readlane_b32", self.s_t[1], t[0], 16),
                    I("v_mov_b32", t[1], self.s_t[1]), I("v_add_f32", t[0], self.s_t[0], t[1]), I("v_rcp_f32", X[1], t[0]), s_nop(0)]
            # every column of O_solo^T is the same query: lanes l31 == 0 (hi = 0, 1) store 8 bytes per group
            out += [I("v_cmp_eq_u32", VCC, 0, self.vL31), I("s_and_saveexec_b64", self.s_exec, VCC, xdef=[EXEC])]
            out += [I("v_lshlrev_b32", X[0], 3, self.vHI)]                                           # hi * 8 bytes
            out += [I("s_lshl_b32", self.s_t[2], self.s_outrow, 8), I("s_add_u32", self.s_t[2], self.s_t[2], self.s_obase)]   # row 256
            for db in range(2):
                for g in range(4):
                    src = self.aOS.sub(16 * db + 4 * g, 4)
                    out += [I("v_accvgpr_read_b32", X[8 + k], src[k]) for k in range(4)]
                    out += [I("v_mul_f32", X[8 + k], X[8 + k], X[1]) for k in range(4)]
                    out += [I("v_cvt_pk_f16_f32", X[2], X[8], X[9]), I("v_cvt_pk_f16_f32", X[3], X[10], X[11]),
                            I("buffer_store_dwordx2", X[2:4], X[0], self.s_ro, self.s_t[2], ndef=0, mods=f"offen offset:{2 * (32 * db + 8 * g)}"), s_nop(1)]
            out += [I("s_mov_b64", EXEC, self.s_exec)]
        return fix_hazards(out, "epilogue_s" if solo else "epilogue_r", self.report)

    def stamp(self, k):
        if not self.prof: return []
        return [s_waitcnt(lgkm=0), I("s_memtime", self.s_tm), s_waitcnt(lgkm=0),
                I("s_sub_u32", self.s_t[7], self.s_tm[0], self.s_t0), I("s_mov_b32", self.s_t0, self.s_tm[0]),
                I("s_mov_b64", EXEC, 1 << k), I("v_add_u32", self.vPACC, self.s_t[7], self.vPACC), I("s_mov_b64", EXEC, -1)]

    def item_offsets(self, P, item, dst):
        """dst = byte offset of the item's first row + h * 128 in qkv (b = item / H, h = item % H); t[0], t[1] clobbered; returns (b in t0, h in t1)."""
        add = P.add
        t = self.s_t
        st = self.new_site()
        add(I("s_cmp_eq_u32", self.s_H, 1, ndef=0)); add(I("s_cbranch_scc1", f"L_h1_{st}", ndef=0))
        add(I("s_mul_hi_u32", t[0], item, self.s_magh)); add(I("s_mul_i32", t[1], t[0], self.s_H)); add(I("s_sub_u32", t[1], item, t[1]))
        add(I("s_branch", f"L_hd_{st}", ndef=0))
        P.label(f"L_h1_{st}"); add(I("s_mov_b32", t[0], item)); add(I("s_mov_b32", t[1], 0))
        P.label(f"L_hd_{st}")
        add(I("s_mul_i32", t[2], t[0], 257)); add(I("s_mul_i32", dst, t[2], self.s_ld2)); add(I("s_lshl_b32", t[3], t[1], 7)); add(I("s_add_u32", dst, dst, t[3]))

    def q_prefetch(self, base):
        out = []
        for q in range(3):
            for ks in range(4):
                out.append(I("buffer_load_dwordx4", self.vQN.sub(16 * q + 4 * ks, 4), self.vQOFF[q], self.s_rq, base, mods=f"offen offset:{32 * ks}"))
        return out

    def build(self):
        P = Program()
        add = P.add
        t = self.s_t
        X = self.vX
        add(I("s_load_dwordx8", self.s_args.sub(0, 8), self.s_karg, 0x0))
        add(I("s_load_dwordx4", self.s_args.sub(8, 4), self.s_karg, 0x20))
        if self.prof: add(I("s_load_dwordx2", self.s_prof, self.s_karg, 0x30)); add(I("v_mov_b32", self.vPACC, 0))
        add(I("v_and_b32", self.vLANE, 63, self.v_tid)); add(I("v_lshrrev_b32", X[0], 6, self.v_tid))
        add(I("v_and_b32", self.vL31, 31, self.vLANE)); add(I("v_lshrrev_b32", self.vHI, 5, self.vLANE))
        add(s_nop(0)); add(I("v_readfirstlane_b32", self.s_w, X[0]))
        add(s_waitcnt(lgkm=0))
        if self.prof: add(I("s_memtime", self.s_tm)); add(s_waitcnt(lgkm=0)); add(I("s_mov_b32", self.s_t0, self.s_tm[0]))
        add(I("s_cmp_ge_u32", self.s_wg, self.s_nitems, ndef=0)); add(I("s_cbranch_scc1", "L_exit", ndef=0))
        add(I("s_lshl_b32", self.s_ld2, self.s_ld, 1)); add(I("s_lshl_b32", self.s_outrow, self.s_H, 7)); add(I("s_mov_b32", self.s_kcol, self.s_outrow))
        for r, base, nbytes in ((self.s_rq, self.s_qkv, self.s_qbytes), (self.s_ro, self.s_out, self.s_obytes)):
            add(I("s_mov_b32", r[0], base[0])); add(I("s_and_b32", r[1], base[1], 0xffff)); add(I("s_mov_b32", r[2], nbytes)); add(I("s_mov_b32", r[3], 0x00020000))
        add(I("v_mov_b32", self.vNINF, NEG_INF)); add(I("s_mov_b32", self.s_ones, 0x3C003C00))
        if self.lazy_max: add(I("v_rcp_f32", self.vRSC, self.s_scale))
        # K fragment addresses (attention.hip tile_addr<64>): row l31, chunk (2 ks + hi) ^ swz64(l31), swz64(r) = (((r >> 1) & 1) << 2) | ((r >> 2) & 3)
        add(I("v_bfe_u32", X[0], self.vL31, 1, 1)); add(I("v_bfe_u32", X[1], self.vL31, 2, 2)); add(I("v_lshl_or_b32", X[0], X[0], 2, X[1]))
        for ks in range(4):
            add(I("v_or_b32", X[1], 2 * ks, self.vHI)); add(I("v_xor_b32", X[1], X[1], X[0])); add(I("v_lshlrev_b32", X[1], 4, X[1]))
            add(I("v_lshl_add_u32", self.vKADDR[ks], self.vL31, 7, X[1]))
        # V tr-read addresses (attention.hip v_off at DH = 64): r = 4 hi + (p16 >> 2) + 8 hf, c = 4 db + 2 g16 + ((p16 & 3) >> 1)
        p16, g16 = X[1], X[2]
        add(I("v_and_b32", p16, 15, self.vLANE)); add(I("v_lshrrev_b32", g16, 4, self.vL31))
        add(I("v_lshrrev_b32", X[3], 2, p16))                                                   # p16 >> 2
        add(I("v_bfe_u32", X[4], p16, 3, 1))                                                    # (r >> 1) & 1 = (p16 >> 3) & 1
        add(I("v_bfe_u32", X[5], p16, 1, 1))                                                    # (p16 & 3) >> 1
        add(I("v_and_b32", X[6], 1, p16))                                                       # p16 & 1
        for db in range(2):
            for hf in range(2):
                d = self.vVADDR[db][hf]
                add(I("v_lshl_add_u32", X[7], self.vHI, 2, X[3])); add(I("v_add_u32", X[7], 8 * hf, X[7]))     # r
                add(I("v_add_u32", X[8], 2 * hf, self.vHI)); add(I("v_and_b32", X[8], 3, X[8]))                 # (r >> 2) & 3
                add(I("v_lshl_or_b32", X[8], X[4], 2, X[8]))                                                    # swz64(r)
                add(I("v_lshl_or_b32", X[9], g16, 1, X[5])); add(I("v_or_b32", X[9], 4 * db, X[9]))             # c
                add(I("v_xor_b32", X[9], X[9], X[8])); add(I("v_lshlrev_b32", X[9], 4, X[9]))
                add(I("v_lshl_add_u32", X[9], X[7], 7, X[9])); add(I("v_lshl_add_u32", X[9], X[6], 3, X[9]))
                add(I("v_add_u32", d, KBYTES, X[9]))
        # strip: write at key * 2 (lane & 31), read at (4 hi) * 2 (+ 32 sp, + 16 for the second run)
        add(I("v_lshlrev_b32", self.vSTRIP[0], 1, self.vL31)); add(I("v_add_u32", self.vSTRIP[0], STRIP, self.vSTRIP[0]))
        add(I("v_lshlrev_b32", self.vSTRIP[1], 3, self.vHI)); add(I("v_add_u32", self.vSTRIP[1], STRIP, self.vSTRIP[1]))
        if self.split:
            # class row, shared form: per-wave addresses — the code is the same for the four waves
            add(I("s_lshl_b32", t[0], self.s_w, 12))
            for ks in range(4): add(I("v_add_u32", self.vKSADDR[ks], t[0], self.vKADDR[ks]))                   # key step w
            add(I("s_lshr_b32", t[1], self.s_w, 1)); add(I("s_lshl_b32", t[1], t[1], 11))                      # sp_w * 16 rows * 128 B
            add(I("s_and_b32", t[0], self.s_w, 1)); add(I("s_cmp_eq_u32", t[0], 1, ndef=0)); add(I("s_cselect_b64", VCC, -1, 0))
            for hf in range(2):
                add(I("v_cndmask_b32", self.vVSADDR[hf], self.vVADDR[0][hf], self.vVADDR[1][hf], VCC))          # db_w
                add(I("v_add_u32", self.vVSADDR[hf], t[1], self.vVSADDR[hf]))
            add(I("s_lshl_b32", t[2], self.s_w, 7)); add(I("v_lshl_add_u32", self.vSWR, self.vL31, 2, t[2])); add(I("v_add_u32", self.vSWR, SSTRIP, self.vSWR))
            add(I("v_lshlrev_b32", self.vSRD, 2, self.vL31)); add(I("v_add_u32", self.vSRD, SSTRIP, self.vSRD))
            add(I("v_lshlrev_b32", self.vPWR, 1, self.vL31)); add(I("v_add_u32", self.vPWR, PSTRIP, self.vPWR))
            add(I("s_lshr_b32", t[1], self.s_w, 1)); add(I("s_lshl_b32", t[1], t[1], 5))                       # 32 sp_w
            add(I("v_lshl_add_u32", self.vPRD, self.vHI, 3, t[1])); add(I("v_add_u32", self.vPRD, PSTRIP, self.vPRD))
            add(I("v_lshl_add_u32", self.vOWR, self.vHI, 6, t[2])); add(I("v_add_u32", self.vOWR, OSTRIP, self.vOWR))
            add(I("s_lshl_b32", t[0], t[0], 6)); add(I("v_lshl_add_u32", self.vOSOFF, self.vHI, 3, t[0]))      # 8 hi + 64 db_w
            add(I("s_mov_b32", self.s_havep, 0)); add(I("s_mov_b32", self.s_opar, 0)); add(I("s_mov_b32", self.s_oprev, 0))
        # DMA source offsets.  K piece: row = 32 t + (L & 31), chunk 2 ks + (L >> 5).  V piece pi: row 8 pi + (L >> 3), chunk (L & 7) ^ swz64(row)
        add(I("v_mul_lo_u32", X[0], self.vL31, self.s_ld2)); add(I("v_lshl_add_u32", self.vKOFF, self.vHI, 4, X[0]))
        add(I("v_lshrrev_b32", X[0], 3, self.vLANE)); add(I("v_mul_lo_u32", X[1], X[0], self.s_ld2))
        add(I("v_bfe_u32", X[2], self.vLANE, 4, 1)); add(I("v_bfe_u32", X[3], self.vLANE, 5, 1))
        add(I("s_and_b32", t[0], self.s_w, 1)); add(I("s_lshl_b32", t[0], t[0], 1))             # (pi & 1) << 1 with pi = w + 4 k
        add(I("v_lshl_or_b32", X[2], X[2], 2, X[3])); add(I("v_or_b32", X[2], t[0], X[2]))      # swz64(row)
        add(I("v_and_b32", X[3], 7, self.vLANE)); add(I("v_xor_b32", X[3], X[3], X[2])); add(I("v_lshl_add_u32", self.vVOFF, X[3], 4, X[1]))
        add(I("v_lshlrev_b32", self.vVOFFT, 4, X[3])); add(I("v_lshlrev_b32", self.vKOFFT, 4, self.vHI))
        # per-lane row offsets of Q (three rows sets) and O
        for q, row0 in ((0, 0), (1, 128)):
            add(I("s_lshl_b32", t[0], self.s_w, 5)); add(I("s_add_u32", t[0], t[0], row0))
            add(I("v_add_u32", X[0], t[0], self.vL31)); add(I("v_mul_lo_u32", X[1], X[0], self.s_ld2)); add(I("v_lshl_add_u32", self.vQOFF[q], self.vHI, 4, X[1]))
            # O: lower lanes 16 bytes at dims 8k.., upper lanes + 16 bytes
            add(I("v_mul_lo_u32", X[1], X[0], self.s_outrow)); add(I("v_lshl_add_u32", self.vOOFF[q], self.vHI, 4, X[1]))
        add(I("s_lshl_b32", t[0], self.s_ld2, 8)); add(I("v_lshlrev_b32", X[1], 4, self.vHI)); add(I("v_add_u32", self.vQOFF[2], t[0], X[1]))
        # first item: K / V and Q now
        add(I("s_mov_b32", self.s_item, self.s_wg)); add(I("s_mov_b32", self.s_buf, 0)); add(I("s_mov_b32", self.s_nbuf, 0)); add(I("s_mov_b32", self.s_delta, BUF))
        self.item_offsets(P, self.s_item, self.s_qbase)
        add(self.dma_setup(self.s_qbase))
        for g in self.dma_groups(): add(g)
        add(self.q_prefetch(self.s_qbase))
        add(I("s_mov_b32", self.s_nbuf, BUF))
        # ================================================================== item loop
        P.label("L_item")
        add(s_waitcnt(vm=0)); add(I("s_barrier", ndef=0))
        add(self.stamp(0))
        if self.split:
            add(self.solo_merge("top"))
            add(I("s_xor_b32", self.s_opar, self.s_opar, 512))
        # out offset of this item: (b * 257) * outrow + h * 128
        self.item_offsets(P, self.s_item, self.s_qbase)
        add(I("s_mul_i32", self.s_obase, t[2], self.s_outrow)); add(I("s_add_u32", self.s_obase, self.s_obase, t[3]))
        for q in range(2):
            for i in range(16): add(I("v_accvgpr_mov_b32", self.aQ[q][i], self.vQN[16 * q + i]))
        for i in range(16): add(I("v_accvgpr_mov_b32", self.aQS[i], self.vQN[32 + i]))
        for q in range(2):
            for i in range(32): add(I("v_accvgpr_write_b32", self.aO[q][i], 0))
        for i in range(self.aOS.n): add(I("v_accvgpr_write_b32", self.aOS[i], 0))
        for q in range(3): add(I("v_mov_b32", self.vL[q], 0)); add(I("v_mov_b32", self.vALPHA[q], 1.0))
        add(I("s_mov_b32", self.s_pend, 0))
        # the next item (K / V into the other buffer during this item, Q into vQN); past the end: loads that fetch nothing
        add(I("s_add_u32", self.s_next, self.s_item, self.s_G))
        add(I("s_cmp_lt_u32", self.s_next, self.s_nitems, ndef=0)); add(I("s_cselect_b32", self.s_rq[2], self.s_qbytes, 0))
        add(I("s_cselect_b32", t[4], self.s_next, self.s_item))
        self.item_offsets(P, t[4], self.s_nbase)
        add(self.dma_setup(self.s_nbase))
        add(self.q_prefetch(self.s_nbase))
        if self.split:
            self.item_body(P, solo=False)
            add(self.stamp(1))
            add(self.solo_partial_write())
            add(self.epilogue(False))
            add(I("s_mov_b32", self.s_oprev, self.s_obase)); add(I("s_mov_b32", self.s_havep, 1))
        else:
            add(I("s_cmp_eq_u32", self.s_w, 3, ndef=0)); add(I("s_cbranch_scc1", "L_item_solo", ndef=0))
            self.item_body(P, solo=False)
            add(self.stamp(1))
            add(self.epilogue(False))
            add(I("s_branch", "L_item_end", ndef=0))
            P.label("L_item_solo")
            self.item_body(P, solo=True)
            add(self.stamp(1))
            add(self.epilogue(True))
        P.label("L_item_end")
        add(self.stamp(2))
        # swap buffers: fragment addresses move to the other buffer
        for ks in range(4): add(I("v_add_u32", self.vKADDR[ks], self.s_delta, self.vKADDR[ks]))
        for db in range(2):
            for hf in range(2): add(I("v_add_u32", self.vVADDR[db][hf], self.s_delta, self.vVADDR[db][hf]))
        if self.split:
            for ks in range(4): add(I("v_add_u32", self.vKSADDR[ks], self.s_delta, self.vKSADDR[ks]))
            for hf in range(2): add(I("v_add_u32", self.vVSADDR[hf], self.s_delta, self.vVSADDR[hf]))
        add(I("s_sub_u32", self.s_delta, 0, self.s_delta))
        add(I("s_mov_b32", t[0], self.s_buf)); add(I("s_mov_b32", self.s_buf, self.s_nbuf)); add(I("s_mov_b32", self.s_nbuf, t[0]))
        add(I("s_mov_b32", self.s_item, self.s_next))
        add(I("s_cmp_lt_u32", self.s_item, self.s_nitems, ndef=0)); add(I("s_cbranch_scc1", "L_item", ndef=0))
        if self.split:                                 # the last item's class row
            add(s_waitcnt(lgkm=0)); add(I("s_barrier", ndef=0))
            add(self.solo_merge("end"))
        if self.prof:
            add(s_waitcnt(vm=0))
            add(I("s_cmp_eq_u64", self.s_prof, 0, ndef=0)); add(I("s_cbranch_scc1", "L_exit", ndef=0))
            add(I("s_lshl_b32", t[4], self.s_wg, 2)); add(I("s_add_u32", t[4], t[4], self.s_w)); add(I("s_lshl_b32", t[4], t[4], 5))
            add(I("v_lshlrev_b32", X[8], 2, self.vLANE)); add(I("v_add_u32", X[8], t[4], X[8]))
            add(I("s_mov_b64", EXEC, 0xff)); add(I("global_store_dword", X[8], self.vPACC, self.s_prof, ndef=0)); add(I("s_mov_b64", EXEC, -1))
        P.label("L_exit")
        add(s_waitcnt(vm=0))
        add(I("s_endpgm", ndef=0))
        add(self.ool)
        return P


def generate(abl=(), prof=False, name=KNAME, dma_steps=5, solo_split=False, dot_sum=False, lazy_max=True):
    g = Gen(abl=abl, prof=prof, dma_steps=dma_steps, solo_split=solo_split, dot_sum=dot_sum, lazy_max=lazy_max)
    prog = g.build()
    items = fix_hazards(prog.items, "whole", g.report)
    hdr = ("// GENERATED by csrc/asm/attn_clip_d64.py — do not edit.  Register map:\n" + g.S.listing() + "\n" + g.V.listing() + "\n" + g.A.listing()
           + "\n// hazard pads inserted: " + ", ".join(f"{t}={n}" for t, n in g.report))
    n_mfma = sum(1 for x in items if isinstance(x, Ins) and x.is_mfma)
    args = ARGS + ([(8, 48, "global_buffer")] if prof else [])
    text = kernel_text(name, render(items), lds_bytes=LDS_BYTES, kernarg_bytes=KERNARG_BYTES + (8 if prof else 0), n_arch_vgpr=g.V.next,
                       n_acc_vgpr=g.A.next, n_sgpr=g.S.next + 2, args_meta=args, header_comment=hdr)
    return text, g, n_mfma


DEV_VARIANTS = [("prof", dict(prof=True)), ("nosm", dict(prof=True, abl=("nosm",))), ("nodma", dict(prof=True, abl=("nodma",))),
                ("nolds", dict(prof=True, abl=("nolds",))), ("nomfma", dict(prof=True, abl=("nomfma",))),
                ("dma1", dict(prof=True, dma_steps=1)), ("dma2", dict(prof=True, dma_steps=2)), ("dma5", dict(prof=True, dma_steps=3)),
                ("dma9", dict(prof=True, dma_steps=9)), ("split", dict(prof=True, solo_split=True)), ("dotsum", dict(prof=True, dot_sum=True)), ("eagermax", dict(prof=True, lazy_max=False))]


def unique_labels(text, suffix):
    return re.sub(r"\bL_([A-Za-z0-9_]+)", lambda m: f"L{suffix}_{m.group(1)}", text)


if __name__ == "__main__":
    text, g, n = generate()
    if "--dev" in sys.argv:
        parts = [text]
        for i, (vn, kw) in enumerate(DEV_VARIANTS):
            vt, _, _ = generate(name=f"{KNAME}_{vn}", **kw)
            vt = unique_labels(vt, f"c{i}")
            vt = "\n".join(l for l in vt.split("\n") if not l.startswith(".amdgcn_target") and not l.startswith(".amdhsa_code_object_version"))
            parts.append(vt)
        text = "\n".join(parts)
    out = [a for a in sys.argv[1:] if not a.startswith("--")]
    if out:
        with open(out[0], "w") as f: f.write(text)
    else:
        sys.stdout.write(text)
    sys.stderr.write(f"attn_clip_d64: {n} MFMA, arch VGPRs {g.V.next}, AccVGPRs {g.A.next}, SGPRs {g.S.next}; pads: {g.report}\n")
