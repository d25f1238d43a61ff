"""Round-6 GPU tests.

  * the assembly attention kernels at the PRODUCTION queue sizes (VERDICT r5 "What's weak": the suite stopped at 24 work items; the
    persistent queue with more items than CUs and the cross-item prefetch were only reached through the full-depth tests);
  * the prefetcher fed WITHOUT a per-step synchronisation, through a staging-buffer regrow (the device staging
    buffer must not come from the compute stream's pool);
  * the assembly kernels' preconditions: a call they cannot take runs on the C++ kernel instead of failing.

Reference arithmetic: llava/train/llama_flash_attn_monkey_patch.py:81-94 (causal flash attention, padded and varlen), transformers
CLIPAttention (non-causal, dh = 64)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _ref_attn_sample(x, H, dh):
    """x: [S, 3, H, dh] fp32 of ONE sample -> (O [S, H*dh], lse [H, S]); per-sample so that B = 16 / S = 2048 fits."""
    S = x.shape[0]
    q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    s = s.masked_fill(~torch.ones(S, S, device=x.device, dtype=torch.bool).tril(), float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(S, H * dh), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,S,H", [(16, 867, 32), (4, 2048, 40)])
def test_attn_fwd_asm_production_queue_vs_oracle_and_cpp(B, S, H):
    """BASELINE config 3's and config 4's attention calls: 2048 resp. 1280 work items on 256 persistent workgroups (8 resp. 5 items
    per workgroup through the per-XCD queue, every switch with the cross-item prefetch)."""
    from hvlm import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(S + H)
    qkv = (torch.randn(B * S, 3 * H * 128, device="cuda", generator=g) * 0.7).to(BF)
    try:
        _lib.configure("attn_fwd_asm", 0); o0, l0 = ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5)
        _lib.configure("attn_fwd_asm", 1); o1, l1 = ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5)
    finally:
        _lib.configure("attn_fwd_asm", 1)
    assert torch.isfinite(o1.float()).all()
    e0 = e1 = el = 0.0
    for b in range(B):
        orf, lrf = _ref_attn_sample(qkv[b * S:(b + 1) * S].float().view(S, 3, H, 128), H, 128)
        e0 = max(e0, (o0[b * S:(b + 1) * S].float() - orf).abs().max().item())
        e1 = max(e1, (o1[b * S:(b + 1) * S].float() - orf).abs().max().item())
        el = max(el, (l1[b] - lrf).abs().max().item())
    assert e1 <= max(1.5 * e0, 4e-3), (e1, e0)          # tolerance: 4e-3 abs on O(1) bf16 outputs = the C++ kernel's bar (test_gpu_r5)
    assert el <= 2e-3


def test_attn_fwd_asm_production_varlen_vs_oracle():
    """The ragged bench batch's lengths [867, 862, 830, 747] x H = 32, packed rows (cu_seqlens): every sample against its own fp32
    reference, and packed == padded bit for bit."""
    from hvlm import _lib, ops
    lens, H = [867, 862, 830, 747], 32
    Sm = max(lens)
    g = torch.Generator(device="cuda").manual_seed(5)
    full = (torch.randn(len(lens) * Sm, 3 * H * 128, device="cuda", generator=g) * 0.7).to(BF)
    rows = torch.cat([torch.arange(b * Sm, b * Sm + n, device="cuda") for b, n in enumerate(lens)])
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    _lib.configure("attn_fwd_asm", 1)
    op, lp = ops.attn_fwd(full[rows].contiguous(), len(lens), Sm, H, 128, True, 128 ** -0.5, cu_seqlens=cu)
    of, lf = ops.attn_fwd(full, len(lens), Sm, H, 128, True, 128 ** -0.5)
    assert torch.equal(op, of[rows])
    off = 0
    for b, n in enumerate(lens):
        orf, lrf = _ref_attn_sample(full[b * Sm:b * Sm + n].float().view(n, 3, H, 128), H, 128)
        # tolerance: 4e-3 abs + 2^-8 relative, elementwise — a sample's first rows attend to a handful of keys, so |O| reaches 2 - 3 there
        # and ONE bf16 rounding of such an output is already 2^-9 relative (3.9e-3 .. 7.8e-3 abs); the P rounding adds as much again
        err = (op[off:off + n].float() - orf).abs() - 2.0 ** -8 * orf.abs()
        assert err.max().item() <= 4e-3, err.max().item()
        assert (lp[b, :, :n] - lrf).abs().max().item() <= 2e-3
        assert torch.equal(lp[b, :, :n], lf[b, :, :n])
        off += n


def test_clip_attn_asm_production_queue_vs_oracle_and_cpp():
    """One tower pass of the SFT step: 1600 frames x 16 heads = 25600 items on 256 persistent workgroups."""
    from hvlm import _lib, ops
    F16 = torch.float16
    B, H = 1600, 16
    g = torch.Generator(device="cuda").manual_seed(1600)
    qkv = torch.randn(B * 257, 3 * H * 64, device="cuda", generator=g).to(F16)
    try:
        _lib.configure("attn_clip_asm", 0); o0 = ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5)
        _lib.configure("attn_clip_asm", 1); o1 = ops.attn_fwd_f16(qkv, B, 257, H, 64, 64 ** -0.5)
    finally:
        _lib.configure("attn_clip_asm", 1)
    assert torch.isfinite(o1.float()).all()
    e0 = e1 = 0.0
    for c in range(0, B, 100):                                              # fp32 reference in chunks of 100 frames (0.4 GB of scores)
        x = qkv[c * 257:(c + 100) * 257].float().view(100, 257, 3, H, 64)
        q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
        ref = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, -1) @ v).transpose(1, 2).reshape(100 * 257, H * 64)
        e0 = max(e0, (o0[c * 257:(c + 100) * 257].float() - ref).abs().max().item())
        e1 = max(e1, (o1[c * 257:(c + 100) * 257].float() - ref).abs().max().item())
    assert e1 <= max(2.0 * e0, 2e-3), (e1, e0)                              # tolerance: 2e-3 abs (one fp16 rounding of P and of O), as test_gpu_r5


def test_attn_asm_preconditions_fall_back_to_cpp():
    """A forward workspace that is not 64-byte aligned cannot hold the assembly kernel's queue counters: the call must run (on the C++
    kernel, bit-identical to `attn_fwd_asm = 0`), not fail.  Same for a CLIP call past the assembly kernel's 32-bit byte offsets."""
    from hvlm import _lib, ops
    lib = _lib.load()
    B, S, H = 2, 300, 2
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = (torch.randn(B * S, 3 * H * 128, device="cuda", generator=g) * 0.7).to(BF)
    try:
        _lib.configure("attn_fwd_asm", 0); o0, l0 = ops.attn_fwd(qkv, B, S, H, 128, True, 128 ** -0.5)
    finally:
        _lib.configure("attn_fwd_asm", 1)
    raw = torch.empty(2048, dtype=torch.uint8, device="cuda")
    ws = raw[(16 - raw.data_ptr() % 64) % 64:]                               # 16-byte aligned (the old header's contract), not 64
    assert ws.data_ptr() % 64 == 16
    o = torch.empty_like(o0); lse = torch.empty_like(l0)
    rc = lib.hvlm_attn_fwd(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(lse.data_ptr()),
                           ctypes.c_void_p(ws.data_ptr()), B, S, H, 128, qkv.stride(0), 1, 128 ** -0.5, ops._stream())
    assert rc == 0, lib.hvlm_last_error()
    torch.cuda.synchronize()
    assert torch.equal(o, o0) and torch.equal(lse, l0)
    # CLIP: ld so large that rows * ld * 2 >= 2^32 (2 frames of a 4 M-column row stride; only the first 3 * H * 64 columns are read)
    Bc, Hc = 2, 1
    ld = (1 << 32) // (Bc * 257 * 2) + 64
    ld -= ld % 8
    big = torch.zeros(Bc * 257, ld, dtype=torch.float16, device="cuda")
    big[:, :3 * Hc * 64] = torch.randn(Bc * 257, 3 * Hc * 64, device="cuda", generator=g).to(torch.float16)
    o1 = ops.attn_fwd_f16(big, Bc, 257, Hc, 64, 64 ** -0.5)                 # asm enabled, preconditions not met -> C++ kernel
    try:
        _lib.configure("attn_clip_asm", 0); o2 = ops.attn_fwd_f16(big, Bc, 257, Hc, 64, 64 ** -0.5)
    finally:
        _lib.configure("attn_clip_asm", 1)
    assert torch.equal(o1, o2)


def test_prefetch_without_per_step_sync_and_regrow():
    """The trainer's feed with NO synchronisation between steps (train.py enqueues step i, then stages batch i + 1), through a staging
    regrow (distinct-frame batches, then an untiled one ten times the size): losses and gradients equal the synchronous feed's bit
    for bit.  With device staging taken from the compute stream's pool the regrown buffer could be a block the in-flight step
    still uses; it now comes from the copy stream's pool (hvlm/prefetch.py::_dev_alloc)."""
    from hvlm.prefetch import DevicePrefetcher
    from test_gpu_r5 import _batch, tiny
    m, cfg = tiny(training=True)
    m.config.dedup_frames = True
    batches = [_batch(2, 40 + 3 * i, seed=20 + i, tiled=(i not in (2, 4))) for i in range(6)]

    def run(feed, sync):
        out = []
        for bt in feed:
            torch.manual_seed(7); torch.cuda.manual_seed_all(7)
            o = m(**bt)
            m.backward()
            if sync:
                torch.cuda.synchronize()
            out.append((o.loss.detach().clone(), m.proj_w_g.clone(), m.stack.layers[0].flat_g.clone(), m.embed_g.clone()))
        torch.cuda.synchronize()
        return out

    ref = run([dict(b) for b in batches], True)
    for _ in range(2):                                                       # twice: the second pass starts from a warm allocator
        torch.cuda.empty_cache() if _ == 0 else None
        feed = DevicePrefetcher([dict(b) for b in batches], m.device, dedup_period=10)
        got = run(feed, False)
        assert len(got) == len(ref)
        for (l0, a0, b0, c0), (l1, a1, b1, c1) in zip(ref, got):
            assert float(l0) == float(l1)
            assert torch.equal(a0, a1) and torch.equal(b0, b1) and torch.equal(c0, c1)
    assert feed.tiled_last is True and feed.bytes_last == 2 * 10 * 3 * 224 * 224 * 2


@pytest.mark.parametrize("S0,n", [(62, 4), (1022, 3), (4092, 3), (4200, 2)])
def test_decode_step_lane_group_attention_split_boundaries_vs_oracle(monkeypatch, S0, n):
    """The decode step's lane-group attention (csrc/decode.hip attn_decode_split_kernel, round 6): splits of <= 64 keys around the
    changes of the split count (1 -> 2 splits at 65 keys, 16 -> 17 at 1025, the 32-split cap at 4096 keys with 128-key splits) and the
    fall-back to the round-3 kernel beyond it (S > 4096), 1 layer of 7B width against the fp32 oracle inside the bf16-eager envelope;
    and both kernels against each other at S = 867 (handsonvlm/model/language_model/handsonvlm.py:543-647 decodes at batch 1)."""
    from test_gpu_decode_r3 import _check, _decode_run, _stack
    st, cfg = _stack(4096, 11008, 32, 1, max_pos=4400)
    xfull, rows, got = _decode_run(st, cfg, S0, n, 23 + S0, monkeypatch, False)
    _check(st, cfg, xfull, rows, got, f"lane-group decode attention from S0={S0}")


def test_decode_step_attention_v1_v2_agree(monkeypatch):
    from hvlm import _lib
    from test_gpu_decode_r3 import _decode_run, _stack
    st, cfg = _stack(4096, 11008, 32, 2)
    res = {}
    for v1 in (0, 1):
        _lib.configure("decode_attn_v1", v1)
        try:
            _, rows, got = _decode_run(st, cfg, 867, 6, 31, monkeypatch, False)
        finally:
            _lib.configure("decode_attn_v1", 0)
        res[v1] = got
    d = (res[0] - res[1]).norm() / res[1].norm()
    assert torch.isfinite(res[0]).all() and d < 5e-3, float(d)          # two bf16 roundings of the attention output per layer apart


@pytest.mark.parametrize("M,N,K,res", [(867, 22016, 4096, False), (600, 22016, 4096, True), (867, 5632, 4096, True)])
def test_gemm_column_tail_split_vs_fp32_and_one_launch(monkeypatch, M, N, K, res):
    """ops.gemm_nt's column-panel tail split for few row panels (the batch-1 prefill's gate|up projection: 344 tiles on 256 CUs): body
    columns as one launch + the rest as split-K.  Against the fp32 product (tolerance: 2^-8 relative to the row-wise operand norms — one
    bf16 rounding of the output) and within one bf16 ulp of the one-launch kernel (HVLM_GEMM_TAIL=0), elementwise."""
    from hvlm import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(BF)
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(BF)
    r = (torch.randn(M, N, device="cuda", generator=g)).to(BF) if res else None
    calls = []
    prev = ops._gemm_hook
    ops.set_gemm_hook(lambda *args: calls.append(args))
    try:
        got = ops.gemm_nt(a, b, residual=r)
    finally:
        ops.set_gemm_hook(prev)
    monkeypatch.setattr(ops, "_GEMM_TAIL", False)
    one = ops.gemm_nt(a, b, residual=r)
    ref = a.float() @ b.float().t() + (r.float() if res else 0.0)
    tiles = ((M + 255) // 256) * (N // 256)
    if 256 < tiles < 512:
        assert any(len(c) == 5 and c[4] == 2 for c in calls), calls            # the two-launch path was taken
    err = (got.float() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 2e-2).all(), float(err.max())
    ulp = 2.0 ** (torch.floor(torch.log2(one.float().abs().clamp_min(1e-30))) - 7)
    assert ((got.float() - one.float()).abs() <= ulp + 1e-6).all()


def _engine_err(st, cache):
    """the persistent decode kernel's timeout flag (decode.hip: EngineArgs::err, in the step's workspace behind the combine tickets)"""
    ws = st._decode_tables(cache)[8]
    D, F = st.D, st.F
    off = (4 * D + 2 * F + 64) * 2 + 128 * 8 * 130 * 4 + 128 * 4 + 64
    return int(ws[off:off + 4].view(torch.int32).item())


@pytest.mark.parametrize("D,F,H,L,S0", [(4096, 11008, 32, 2, 867), (4096, 11008, 32, 1, 0), (4096, 11008, 32, 1, 126), (5120, 13824, 40, 1, 600),
                                        (256, 512, 2, 3, 40)])
def test_decode_engine_bit_identical_to_launch_sequence(monkeypatch, D, F, H, L, S0):
    """decode_stack_kernel (one persistent launch for the whole stack, grid barriers between the phases, the next phase's weight rows
    requested before each barrier) against the launch sequence of hvlm_llama_decode_step (5 launches per layer): hidden states of 6 steps
    and the K | V rows they leave in the cache are bit-identical — same per-row summation order, same RMSNorm / SwiGLU / RoPE arithmetic,
    the same attention body; and no barrier timed out.  (handsonvlm/model/language_model/handsonvlm.py:543-647: batch-1 decoding.)"""
    from hvlm import _lib
    from test_gpu_decode_r3 import _stack
    st, cfg = _stack(D, F, H, L)
    n = 6
    g = torch.Generator(device="cuda").manual_seed(5 + S0)
    xfull = torch.randn(S0 + n, D, device="cuda", generator=g).to(BF)
    monkeypatch.setenv("HVLM_DECODE_GRAPH", "0")
    res = {}
    for eng in (0, 1, 0, 1):
        _lib.configure("decode_engine", eng)
        try:
            st._kv_cache = None; st._dec_tables = None
            cache = st.new_cache(S0 + n + 1)
            for c in cache:
                c.copy_(torch.full(c.shape, float("nan"), device="cuda", dtype=BF))
            if S0 > 0:
                st.forward(xfull[:S0].contiguous(), 1, S0, save=False, cache=cache)
            hs = [st.decode_step(xfull[S0 + i:S0 + i + 1].contiguous(), S0 + i, cache).view(-1).clone() for i in range(n)]
            torch.cuda.synchronize()
            rows = [c[S0:S0 + n, D:].clone() for c in cache]
            if eng:
                assert _engine_err(st, cache) == 0, "a grid barrier of the persistent decode kernel timed out"
        finally:
            _lib.configure("decode_engine", 0)
        if eng in res:
            for a, b in zip(res[eng][0], hs):
                assert torch.equal(a, b)                                  # run-to-run identical
        res[eng] = (hs, rows)
    for i, (a, b) in enumerate(zip(res[0][0], res[1][0])):
        assert torch.isfinite(b.float()).all(), i
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_decode_attention_handoff_under_background_load(monkeypatch):
    """The split attention's cross-workgroup hand-off (agent-scope partial stores, a ticket per head, the last workgroup's agent-scope loads) is
    only as good as its ordering under UNEVEN load: the same 24 decode steps of a 2-layer 7B-width stack, once on an idle chip and once beside a
    side stream that keeps HBM busy with 256 MB copies, must be bit-identical (a stale or torn partial would not be)."""
    from test_gpu_decode_r3 import _stack
    st, cfg = _stack(4096, 11008, 32, 2)
    D, S0, n = 4096, 867, 24
    g = torch.Generator(device="cuda").manual_seed(77)
    xfull = torch.randn(S0 + n, D, device="cuda", generator=g).to(BF)
    monkeypatch.setenv("HVLM_DECODE_GRAPH", "0")
    side = torch.cuda.Stream()
    a = torch.empty(128 * 2 ** 20, dtype=torch.int16, device="cuda"); b = torch.empty_like(a)
    res = []
    for load in (False, True, True):
        st._kv_cache = None; st._dec_tables = None
        cache = st.new_cache(S0 + n + 1)
        st.forward(xfull[:S0].contiguous(), 1, S0, save=False, cache=cache)
        torch.cuda.synchronize()
        hs = []
        for i in range(n):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        b.copy_(a, non_blocking=True)
            hs.append(st.decode_step(xfull[S0 + i:S0 + i + 1].contiguous(), S0 + i, cache).view(-1).clone())
        torch.cuda.synchronize()
        res.append(torch.stack(hs))
    assert torch.isfinite(res[0].float()).all()
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
