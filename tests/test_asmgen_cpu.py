"""CPU tests of the assembly generator behind the hand-scheduled attention kernels (handsonvlm-release_amd/csrc/asm): the hazard pass
inserts exactly the wait states measured on hipcc output (profiles/r05a_hazard_probe.txt), the LDS tracker derives the counted
lgkmcnt waits, the interleaver keeps every stream in order, and both kernel generators emit text that assembles for gfx950 inside
the register / LDS budget of one wave per SIMD.  No GPU: clang only assembles."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "handsonvlm-release_amd", "csrc", "asm")
sys.path.insert(0, ASM)

import asmgen as ag  # noqa: E402
from asmgen import I, Filler, Reg, RegFile, add_lgkm_waits, fix_hazards, interleave, mfma_bf16, s_nop  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang"


def _v(i, n=1): return Reg("v", i, n)
def _a(i, n=1): return Reg("a", i, n)
def _s(i, n=1): return Reg("s", i, n)


def _states_between(items, first, second):
    """issue slots (instructions + s_nop states) strictly between two instructions of a list"""
    i0, i1 = items.index(first), items.index(second)
    return sum(x.states for x in items[i0 + 1:i1])


def test_hazard_pass_inserts_the_measured_wait_states():
    mf = mfma_bf16(_a(0, 16), _a(64, 4), _a(68, 4), 0)
    rd = I("v_accvgpr_read_b32", _v(0), _a(3))
    out = fix_hazards([mf, rd])
    assert _states_between(out, mf, rd) == ag.MFMA_RESULT_STATES == 12
    # an accumulate chain on the same block needs nothing, neither does overwriting the block with C = 0
    chain = mfma_bf16(_a(0, 16), _a(72, 4), _a(76, 4), _a(0, 16))
    over = mfma_bf16(_a(0, 16), _a(72, 4), _a(76, 4), 0)
    assert all(not x.is_nop for x in fix_hazards([mf, chain, over]))
    # an MFMA that reads the block as an OPERAND is a reader like any other
    use = mfma_bf16(_a(16, 16), _a(0, 4), _a(76, 4), 0)
    out = fix_hazards([mf, use])
    assert _states_between(out, mf, use) == 12
    # independent work already between the two counts
    pad = [I("v_add_f32", _v(10 + k), _v(20 + k), _v(30 + k)) for k in range(5)]
    out = fix_hazards([mf] + pad + [rd])
    assert _states_between(out, mf, rd) == 12 and sum(x.states for x in out if x.is_nop) == 7


def test_hazard_pass_valu_rules():
    w = I("v_accvgpr_write_b32", _a(64), _v(1))
    mf = mfma_bf16(_a(0, 16), _a(64, 4), _a(68, 4), 0)
    out = fix_hazards([w, mf])
    assert _states_between(out, w, mf) == ag.VALU_TO_MFMA == 2
    ex = I("v_exp_f32", _v(2), _v(2)); ad = I("v_add_f32", _v(3), _v(2), _v(4))
    assert _states_between(fix_hazards([ex, ad]), ex, ad) == 1
    cv = I("v_cvt_pk_bf16_f32", _v(5), _v(6), _v(7)); mv = I("v_mov_b32", _v(8), _v(5))
    assert _states_between(fix_hazards([cv, mv]), cv, mv) == 1
    mo = I("v_mov_b32", _v(9), _v(10)); pl = I("v_permlane32_swap_b32", _v(9), _v(11), ndef=2)
    assert _states_between(fix_hazards([mo, pl]), mo, pl) == ag.VALU_TO_PERM == 2
    m0 = I("s_mov_b32", ag.M0, _s(4)); dma = I("buffer_load_dwordx4", _v(12), _s(8, 4), _s(5), ndef=0, mods="offen lds", xuse=[ag.M0])
    assert _states_between(fix_hazards([m0, dma]), m0, dma) == 1
    rl = I("v_readfirstlane_b32", _s(6), _v(13)); ld = I("buffer_load_dwordx4", _v(16, 4), _v(14), _s(8, 4), _s(6), mods="offen")
    assert _states_between(fix_hazards([rl, ld]), rl, ld) == ag.VSGPR_TO_VMEM == 5
    # DOT results: accumulation through SrcC of the same opcode is free, a different VALU reading / overwriting the result waits 3 / 4
    d0 = I("v_dot2_f32_f16", _v(40), _v(41), _s(7), 0); d1 = I("v_dot2_f32_f16", _v(40), _v(42), _s(7), _v(40))
    rd2 = I("v_add_f32", _v(43), _v(40), _v(44)); ow = I("v_mov_b32", _v(40), 0)
    assert all(not x.is_nop for x in fix_hazards([d0, d1]))
    assert _states_between(fix_hazards([d1, rd2]), d1, rd2) == ag.DOT_TO_VALU_READ == 3
    assert _states_between(fix_hazards([d1, ow]), d1, ow) == ag.DOT_TO_VALU_WRITE == 4
    # unrelated registers: nothing
    assert all(not x.is_nop for x in fix_hazards([w, mfma_bf16(_a(0, 16), _a(80, 4), _a(84, 4), 0), ex, I("v_add_f32", _v(30), _v(31), _v(32))]))
    # `carry`: the tail of a loop body is checked against its own head without being emitted
    out = fix_hazards([mf], carry=[w])
    assert out[0].is_nop and out[0].states == 2 and out[1] is mf and len(out) == 2


def test_lgkm_tracker_derives_counted_waits():
    r = [I("ds_read_b128", _a(4 * k, 4), _v(0), mods=f"offset:{1024 * k}") for k in range(4)]
    use0 = mfma_bf16(_a(100, 16), _a(0, 4), _a(64, 4), 0)          # needs read 0: three younger reads may stay in flight
    use3 = mfma_bf16(_a(100, 16), _a(12, 4), _a(64, 4), _a(100, 16))
    out, pend = add_lgkm_waits(r + [use0, use3])
    waits = [str(x.ops[0]) for x in out if x.op == "s_waitcnt"]
    assert waits == ["lgkmcnt(3)", "lgkmcnt(0)"] and pend == []
    # a write is counted but defines nothing; overwriting a register with a read still pending waits for that read
    wr = I("ds_write_b32", _v(1), _v(2), ndef=0)
    clobber = I("v_mov_b32", _a(0), 0)
    out, pend = add_lgkm_waits([r[0], wr, clobber])
    assert [str(x.ops[0]) for x in out if x.op == "s_waitcnt"] == ["lgkmcnt(1)"] and len(pend) == 1
    # the counter has four bits: with 20 younger operations the wait saturates at 15 (stronger than needed, never weaker)
    many = [I("ds_read_b64", _v(40 + 2 * k, 2), _v(0), mods=f"offset:{8 * k}") for k in range(21)]
    out, _ = add_lgkm_waits(many + [I("v_mov_b32", _v(100), _v(40))])
    assert [str(x.ops[0]) for x in out if x.op == "s_waitcnt"] == ["lgkmcnt(15)"]
    # an explicit wait in the stream is honoured
    out, pend = add_lgkm_waits([r[0], r[1], ag.s_waitcnt(lgkm=0), use0])
    assert sum(1 for x in out if x.op == "s_waitcnt") == 1 and pend == []


def test_interleave_keeps_orders_and_placement_constraints():
    mf = [mfma_bf16(_a(16 * (k % 4), 16), _a(64, 4), _a(68, 4), _a(16 * (k % 4), 16)) for k in range(8)]
    stream = [Filler([I("v_add_f32", _v(k), _v(k), _v(100))]) for k in range(20)]
    early = [Filler([I("ds_read_b128", _a(80, 4), _v(200), mods="offset:0")], after=2, prio=0),
             Filler([I("ds_read_b128", _a(84, 4), _v(200), mods="offset:16")], after=5, prio=0)]
    out = interleave(mf, early, stream, head_room=2)
    assert [x for x in out if x.is_mfma] == mf
    adds = [x for x in out if x.op == "v_add_f32"]
    assert adds == [f.ins[0] for f in stream]
    assert out.index(adds[0]) > out.index(mf[1])                       # head room: nothing of the stream in the first two gaps
    for f in early:
        assert out.index(f.ins[0]) > out.index(mf[f.after])
    per_gap = [sum(1 for x in out[out.index(mf[k]):out.index(mf[k + 1])] if x.op == "v_add_f32") for k in range(2, 7)]
    assert max(per_gap) - min(per_gap) <= 2                            # spread evenly


def test_register_file_allocator():
    f = RegFile("v", 0, 16)
    a = f.alloc("a", 3); b = f.alloc("pair", 2, 2); c = f.alloc("quad", 4, 4)
    assert (a.idx, b.idx, c.idx) == (0, 4, 8) and str(c) == "v[8:11]" and str(c[1]) == "v9" and str(c.sub(2, 2)) == "v[10:11]"
    with pytest.raises(AssertionError):
        f.alloc("too many", 8)
    with pytest.raises(AssertionError):
        c.sub(2, 4)


def _assemble(text, tmp_path, name):
    src = tmp_path / f"{name}.s"
    src.write_text(text)
    obj = tmp_path / f"{name}.o"
    r = subprocess.run([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]
    return obj


def _check_text(text, lds_limit=160 * 1024):
    labels = re.findall(r"^(L\w+):", text, flags=re.M)
    assert len(labels) == len(set(labels))
    for m in re.finditer(r"lgkmcnt\((\d+)\)", text): assert int(m.group(1)) <= 15
    for m in re.finditer(r"vmcnt\((\d+)\)", text): assert int(m.group(1)) <= 63
    nxt = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", text).group(1))
    acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", text).group(1))
    lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", text).group(1))
    assert nxt <= 512 and acc % 4 == 0 and 4 <= acc <= 256 and lds <= lds_limit
    # every register the body names lies inside what the descriptor declares
    body = text.split(".rodata")[0]
    vmax = max([int(x) for x in re.findall(r"\bv(\d+)\b", body)] + [int(x) for x in re.findall(r"\bv\[\d+:(\d+)\]", body)])
    amax = max([int(x) for x in re.findall(r"\ba(\d+)\b", body)] + [int(x) for x in re.findall(r"\ba\[\d+:(\d+)\]", body)])
    assert vmax < acc and acc + amax < nxt
    return nxt, acc, lds


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not installed")
def test_decoder_forward_generator_assembles_inside_the_budget(tmp_path):
    import attn_fwd_d128 as gen
    text, g, n_mfma = gen.generate()
    nxt, acc, lds = _check_text(text)
    assert g.A.next == 256 and nxt > 480                                # the whole file: one wave per SIMD by construction
    assert "v_mfma_f32_32x32x16_bf16" in text and "buffer_load_dwordx4" in text and " lds" in text and "ds_read_b64_tr_b16" in text
    assert n_mfma == text.count("v_mfma_f32_32x32x16_bf16")
    _assemble(text, tmp_path, "fwd")
    # a clean schedule: the hazard pass found nothing to pad inside the two loop bodies
    loop_pads = [n for tag, n in g.report if tag.split("_")[0] in ("step0", "first", "second")]
    assert len(loop_pads) == 6 and max(loop_pads) <= 4, g.report
    # the alternative K image (the granule image, dev variant `kgran`) is a different, valid program
    text2, g2, _ = gen.generate(k_rows=False)
    assert text2 != text
    _check_text(text2); _assemble(text2, tmp_path, "fwd_kgran")


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not installed")
def test_clip_generator_assembles_inside_the_budget(tmp_path):
    import attn_clip_d64 as gen
    text, g, n_mfma = gen.generate()
    _check_text(text)
    # two bodies (waves 0-2 / wave 3): 16 QK MFMAs in the prologue + 7 x 16 + 2 x 8 in the steps; wave 3 adds 8 per step for the class row
    assert n_mfma == 2 * 144 + 9 * 8 == text.count("v_mfma_f32_32x32x16_f16")
    assert text.count("s_barrier") == 1
    _assemble(text, tmp_path, "clip")
    # the balanced dev variant: one body, + 3 rounds x 4 for the class row's scores and 9 for its P V on every wave
    text2, g2, n2 = gen.generate(solo_split=True)
    assert n2 == 144 + 12 + 9 and text2.count("s_barrier") == 3 and len(text) > 1.7 * len(text2)
    _check_text(text2); _assemble(text2, tmp_path, "clip_split")


@pytest.mark.skipif(shutil.which("make") is None or not os.path.exists(CLANG), reason="toolchain")
def test_code_objects_embedded_in_the_library_match_the_generators():
    """The library embeds the code objects as byte arrays (build/*_co.inc): they are present and name the kernels."""
    build = os.path.join(ROOT, "handsonvlm-release_amd", "build")
    if not os.path.exists(os.path.join(build, "attn_fwd_d128.co")):
        pytest.skip("library not built in this tree")
    for co, kname in (("attn_fwd_d128.co", b"hvlm_fa_fwd_d128_causal"), ("attn_clip_d64.co", b"hvlm_fa_clip_d64_s257")):
        blob = open(os.path.join(build, co), "rb").read()
        assert blob[:4] == b"\x7fELF" and kname in blob and kname + b".kd" in blob
        inc = open(os.path.join(build, co.replace(".co", "_co.inc"))).read()
        body = inc[inc.index("{") + 1:inc.rindex("}")]
        assert bytes(int(x) for x in body.replace("\n", " ").split(",") if x.strip()) == blob
