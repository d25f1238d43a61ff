"""Full-depth parity, the holes the round-4 verdict listed (handsonvlm/model/language_model/handsonvlm.py:119-200):

  (g) 13B, L = 40, S = 2048 (BASELINE config 4): forward AND backward against the fp32 oracle — loss, final hidden state and the
      gradients of layers.0 q|k|v, layers.39 down_proj, the final norm, the projector and the <hand_traj> embedding row — with the
      oracle's own bf16-eager run as the envelope (tests/test_gpu_fulldepth_r4.py did the forward only; config 4's bench child runs
      a backward at this depth that nothing had compared with anything);
  (h) 7B, L = 32, a RAGGED batch of four samples (pad tails 0 / 37 / 5 / 120 tokens), padded rows and packed rows
      (config.pack_sequences), loss + the same gradient set against the oracle (the round-4 fixture was B = 1, where "packed" only
      drops 5 pad rows);
  (i) BASELINE config 2's geometry: 336 px frames (24 x 24 patches -> 100 + 576 visual tokens), T = 512 -> S = 1187, L = 32 forward
      against the oracle, then the KV-cached decode path: prefill + 6 single-token steps, every step vs the oracle's row.

The machinery (trained-like weights with outlier channels, the oracle passes, the bf16-eager envelope: HIP <= 1.25 x the oracle's own
bf16-eager distance + 1e-3) is test_gpu_fulldepth_r4's.  Host memory: the fp32 oracle backward keeps ~1.8 GB of activations per
layer (S = 2048: attention probabilities 0.67 GB per layer alone) next to 52 GB of fp32 weights — tests skip on hosts with less
free memory than they need instead of taking the box down.  Numbers go to gpurun_out/fulldepth_r5_report.json."""
import gc
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle import synth  # noqa: E402
from test_gpu_fulldepth_r4 import (BF, GRAD_KEYS, OUTLIER_CH, OUTLIER_GAIN, _batch, _build, _hip_grads, _hip_pass, _oracle_pass,  # noqa: E402
                                   _threads, rel)

REPORT = {}


def _mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


SKIPPED = {}       # test name -> reason: written into the report and checked by test_zz_no_silent_fulldepth_skip


def _need_host_gb(gb, name):
    have = _mem_available_gb()
    if have < gb:
        why = f"the fp32 oracle of this test needs ~{gb} GB of host memory, {have:.0f} GB available"
        SKIPPED[name] = why
        REPORT.setdefault("skipped", {})[name] = why
        _write()
        pytest.skip(why)


# g_embed_hand is ONE 4096-element row = the sum of the 4 B hand-token rows' gradients: its distance from fp32 is decided by which handful
# of bf16 roundings flip, and two bf16 runs of the same function differ from EACH OTHER by as much as either differs from fp32
# (measured at B = 4: HIP padded vs HIP packed 6.9e-2, HIP padded vs fp32 8.3e-2, HIP packed vs fp32 5.3e-2, bf16-eager oracle vs fp32
# 6.4e-2).  It gets 1.5 x the envelope; every large tensor keeps 1.25 x.
ENVELOPE_FACTOR = {"g_embed_hand": 1.5}


def _envelope_asserts(r, keys, cap=8e-2):
    assert r["loss_rel"] <= max(1.25 * r["loss_rel_bf16_eager"], 2e-3), r
    for k in keys:
        h, e = r["hip_vs_fp32_oracle"][k], r["bf16_eager_oracle_vs_fp32_oracle"][k]
        assert h <= ENVELOPE_FACTOR.get(k, 1.25) * e + 1e-3, (k, h, e)
        assert h < cap * (1.25 if k in ENVELOPE_FACTOR else 1.0), (k, h)


def _write():
    REPORT.setdefault("skipped", {})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fulldepth_r5_report.json", "w") as f:
        json.dump(REPORT, f, indent=1)


def test_g_13b_L40_S2048_forward_backward_vs_oracle():
    _need_host_gb(230, "test_g_13b_L40_S2048_forward_backward_vs_oracle")
    gc.collect(); torch.cuda.empty_cache()
    m, cfg, clip_np = _build("13b", training=True)
    bt = _batch(1, 1693, seed=50, pad_tail=(0,))
    cfg.lm_head_labelled_rows, cfg.pack_sequences = False, False
    out = _hip_pass(m, bt, output_hidden_states=True)
    assert out.logits.shape[1] == 2048
    hip = {"loss": float(out.loss.detach()), "hidden": out.hidden_states[-1].float().cpu(),
           "logits_slice": out.logits.float().cpu()[:, ::7, ::251].clone()}
    out.loss.backward()
    torch.cuda.synchronize()
    hip.update(_hip_grads(m, cfg, bt))
    del out
    # the INFERENCE-mode forward on the same weights and batch (no saved activations, the evaluation path: what round 4's test_d ran with its own two oracle
    # passes — 114 s of the suite for the same references)
    m.training = False
    with torch.no_grad():
        out = _hip_pass(m, bt, output_hidden_states=True)
    inf = {"loss": float(out.loss), "hidden": out.hidden_states[-1].float().cpu(), "logits_slice": out.logits.float().cpu()[:, ::7, ::251].clone()}
    m.training = True
    del out
    ref = _oracle_pass(m, cfg, clip_np, bt, torch.float32, torch.float32, backward=True)
    env = _oracle_pass(m, cfg, clip_np, bt, BF, torch.float16, backward=True)
    keys = ("hidden", "logits_slice") + GRAD_KEYS
    r = {"loss_hip": hip["loss"], "loss_oracle_fp32": ref["loss"], "loss_oracle_bf16_eager": env["loss"],
         "loss_rel": abs(hip["loss"] - ref["loss"]) / abs(ref["loss"]), "loss_rel_bf16_eager": abs(env["loss"] - ref["loss"]) / abs(ref["loss"]),
         "hip_vs_fp32_oracle": {k: rel(hip[k], ref[k]) for k in keys},
         "bf16_eager_oracle_vs_fp32_oracle": {k: rel(env[k], ref[k]) for k in keys},
         "oracle_wall_s": {"fp32": ref["wall_s"], "bf16_eager": env["wall_s"]},
         "inference_forward": {"loss_rel": abs(inf["loss"] - ref["loss"]) / abs(ref["loss"]),
                               "hidden": rel(inf["hidden"], ref["hidden"]), "logits_slice": rel(inf["logits_slice"], ref["logits_slice"])}}
    REPORT["g_13b_L40_S2048_fwd_bwd"] = r
    print(json.dumps(r, indent=1))
    _write()
    del m, ref, env
    gc.collect(); torch.cuda.empty_cache()
    _envelope_asserts(r, keys, cap=1.0e-1)          # (L = 40, S = 2048: the bf16-eager oracle itself is 5.8e-2 from fp32 on the hidden state)
    fi = r["inference_forward"]                     # (the assertions of round 4's test_d, on the same references)
    assert fi["loss_rel"] <= max(1.25 * r["loss_rel_bf16_eager"], 2e-3), fi
    for k in ("hidden", "logits_slice"):
        assert fi[k] <= 1.25 * r["bf16_eager_oracle_vs_fp32_oracle"][k] + 1e-3 and fi[k] < 8e-2, (k, fi, r["bf16_eager_oracle_vs_fp32_oracle"][k])


def test_h_7b_L32_ragged_batch_padded_and_packed_vs_oracle():
    _need_host_gb(150, "test_h_7b_L32_ragged_batch_padded_and_packed_vs_oracle")
    gc.collect(); torch.cuda.empty_cache()
    m, cfg, clip_np = _build("7b", training=True)
    pads = (0, 37, 5, 120)
    bt = _batch(4, 512, seed=60, pad_tail=pads)
    B, S = 4, 512 + 355
    valid = torch.ones(B, S, dtype=torch.bool)
    for b, pt in enumerate(pads):
        if pt: valid[b, S - pt:] = False            # pad rows hold whatever the padding produced (the reference masks nothing there either)
    runs = {}
    try:
        for name, pack in (("padded", False), ("packed", True)):
            cfg.lm_head_labelled_rows, cfg.pack_sequences = False, pack
            m.stack.recompute = 0
            out = _hip_pass(m, bt, output_hidden_states=True)
            h = out.hidden_states[-1].float().cpu()
            if pack:                                # [M, D] rows of sample b = cu[b] .. cu[b + 1]
                cu = out.cu_seqlens.cpu().tolist()
                full = torch.zeros(B, S, h.shape[-1])
                for b in range(B):
                    full[b, : cu[b + 1] - cu[b]] = h[cu[b]:cu[b + 1]]
                    assert cu[b + 1] - cu[b] == S - pads[b]
                h = full
            rr = {"loss": float(out.loss.detach()), "hidden": h}
            out.loss.backward()
            torch.cuda.synchronize()
            rr.update(_hip_grads(m, cfg, bt))
            runs[name] = rr
            del out
    finally:
        cfg.pack_sequences = False
    ref = _oracle_pass(m, cfg, clip_np, bt, torch.float32, torch.float32, backward=True)
    env = _oracle_pass(m, cfg, clip_np, bt, BF, torch.float16, backward=True)
    rep = {"pad_tail": list(pads), "oracle_wall_s": {"fp32": ref["wall_s"], "bf16_eager": env["wall_s"]}}
    envq = {"hidden": rel(env["hidden"][valid], ref["hidden"][valid]), **{k: rel(env[k], ref[k]) for k in GRAD_KEYS}}
    for name, rr in runs.items():
        q = {"hidden": rel(rr["hidden"][valid], ref["hidden"][valid]), **{k: rel(rr[k], ref[k]) for k in GRAD_KEYS}}
        rep[name] = {"loss_hip": rr["loss"], "loss_oracle_fp32": ref["loss"], "loss_oracle_bf16_eager": env["loss"],
                     "loss_rel": abs(rr["loss"] - ref["loss"]) / abs(ref["loss"]),
                     "loss_rel_bf16_eager": abs(env["loss"] - ref["loss"]) / abs(ref["loss"]),
                     "hip_vs_fp32_oracle": q, "bf16_eager_oracle_vs_fp32_oracle": envq}
    rep["packed_vs_padded"] = {"loss_rel": abs(runs["packed"]["loss"] - runs["padded"]["loss"]) / abs(runs["padded"]["loss"]),
                               **{k: rel(runs["packed"][k], runs["padded"][k]) for k in GRAD_KEYS}}
    REPORT["h_7b_L32_B4_ragged"] = rep
    print(json.dumps(rep, indent=1))
    _write()
    del m, ref, env
    gc.collect(); torch.cuda.empty_cache()
    for name in ("padded", "packed"):
        _envelope_asserts(rep[name], ("hidden",) + GRAD_KEYS)
    # packed and padded rows run the SAME function on other GEMM tile boundaries (M = 3306 instead of 3468: other rows fall into the
    # split-K tail), so they differ by bf16 noise, not bit for bit: the loss inside the bf16-eager oracle's own distance from fp32
    assert rep["packed_vs_padded"]["loss_rel"] <= max(rep["padded"]["loss_rel_bf16_eager"], 1e-3)


def _build_336(training):
    """test_gpu_fulldepth_r4._build at 336 px: ViT-L/14 with 24 x 24 patches (577 positions)."""
    from hvlm.config import HandsOnVLMConfig
    from hvlm.model import HandsOnVLMForCausalLM
    torch.manual_seed(20251003)
    cfg = HandsOnVLMConfig.vicuna_7b()
    cfg.vision.image_size = 336
    L = int(os.environ["HVLM_FULLDEPTH_L"]) if os.environ.get("HVLM_FULLDEPTH_L") else None
    if L is not None:
        cfg.num_hidden_layers = L
    m = HandsOnVLMForCausalLM(cfg, device="cuda:0", training=training).init_random(seed=0, std=0.02)
    D = cfg.hidden_size
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for lp in m.stack.layers:
            for k in ("ln1", "ln2"):
                lp.w[k].copy_((1.0 + 0.1 * torch.randn(D, device="cuda", generator=g)).to(BF))
        m.stack.norm_w.copy_((1.0 + 0.1 * torch.randn(D, device="cuda", generator=g)).to(BF))
        ch = torch.tensor(OUTLIER_CH, device="cuda")
        for li in (1, 2):
            if li < len(m.stack.layers):
                m.stack.layers[li].w["down"][ch] *= OUTLIER_GAIN
                m.stack.layers[li].refresh_transposes()
        m.proj_b.copy_((0.02 * torch.randn(D, device="cuda", generator=g)).to(BF))
    clip_np = synth.clip_params(dict(clip_d=1024, clip_mlp=4096, clip_layers=24, clip_heads=16, patch=14, image=336), seed=77)
    m.tower.load_state_dict(synth.to_torch(clip_np))
    return m, cfg, clip_np


def test_i_7b_336px_S1187_forward_and_kv_decode_vs_oracle(monkeypatch):
    _need_host_gb(90, "test_i_7b_L32_336px_prefill_and_cached_decode_vs_oracle")
    gc.collect(); torch.cuda.empty_cache()
    from hvlm import ops
    monkeypatch.setenv("HVLM_DECODE_GRAPH", "0")
    m, cfg, clip_np = _build_336(training=False)
    L, H = cfg.num_hidden_layers, cfg.num_attention_heads
    T = 512
    ids, _, mask = synth.make_text_batch(1, T, seed=70, n_hand=(0,), pad_tail=(0,))
    r = np.random.RandomState(71)
    base = torch.from_numpy(r.standard_normal((1, 10, 3, 336, 336)).astype(np.float32)).to(BF)
    images = base[:, torch.arange(100) % 10].contiguous()
    S0, n = 1100, 6
    with torch.no_grad():
        vis = m.visual_tokens(images.cuda())
        assert vis.shape[1] == 100 + 576
        plan = m.build_splice_plan(ids, None, vis.shape[1], 0, True, False)
        assert plan.S == 1187
        x = ops.embed_splice_fwd(m.embed, vis, plan.src, None, None, None, 1, plan.S, m.token_dim, vis.shape[1], 1)
        stk = m.stack
        hid = stk.forward(x, 1, plan.S, save=False).float().cpu()                           # the whole S = 1187 forward
        stk._kv_cache, stk._dec_tables = None, None
        cache = stk.new_cache(S0 + n + 1)
        for c in cache:
            c.copy_(torch.full(c.shape, float("nan"), device="cuda", dtype=BF))
        got = [stk.forward(x[:S0].contiguous(), 1, S0, save=False, cache=cache)[S0 - 1].clone()]
        for t in range(n):
            got.append(stk.decode_step(x[S0 + t:S0 + t + 1].contiguous(), S0 + t, cache).view(-1).clone())
        torch.cuda.synchronize()
        stk._kv_cache, stk._dec_tables = None, None
    got = torch.stack(got).float().cpu()
    rows = [S0 - 1] + [S0 + t for t in range(n)]
    res, visq = {}, {}
    with _threads(), torch.no_grad():
        for name, dt, cdt in (("fp32", torch.float32, torch.float32), ("bf16", BF, torch.float16)):
            cp = synth.to_torch(clip_np, cdt)
            feats10 = R.clip_features(base.flatten(0, 1).to(cdt), cp, 24, 16)                  # [10, 576, 1024]
            feats = feats10.view(1, 10, 576, 1024)[:, torch.arange(100) % 10].to(dt)
            sd = {k: v.detach().to("cpu", dt) for k, v in m.state_dict().items() if not k.startswith("hand_traj_decoder.")}
            tok = torch.nn.functional.linear(feats.reshape(100 * 576, 1024), sd["model.mm_projector.weight"], sd["model.mm_projector.bias"])
            v_or = R.temporal_spatial_pool(tok.view(1, 100, 576, -1))
            xo, _, _ = R.splice_batch(ids, None, mask, sd["model.embed_tokens.weight"], v_or, None, True)
            p = {k: v.detach().to("cpu", dt) for k, v in stk.export_hf("model.").items()}
            res[name] = R.llama_model(xo.to(dt), p, L, H, cfg.rms_norm_eps)[0].float()
            visq[name] = v_or.float()
            del p, sd, cp
            gc.collect()
    rep = {"S": plan.S, "visual_tokens": int(vis.shape[1]),
           "visual_tokens_rel": {"hip": rel(vis.float().cpu(), visq["fp32"]), "bf16_eager": rel(visq["bf16"], visq["fp32"])},
           "final_hidden_rel": {"hip": rel(hid, res["fp32"]), "bf16_eager": rel(res["bf16"], res["fp32"])},
           "decode_rows": rows,
           "decode_hip_vs_fp32_oracle": [rel(got[i], res["fp32"][rows[i]]) for i in range(len(rows))],
           "decode_bf16_eager_vs_fp32_oracle": [rel(res["bf16"][rows[i]], res["fp32"][rows[i]]) for i in range(len(rows))]}
    REPORT["i_7b_L32_336px_S1187"] = rep
    print(json.dumps(rep, indent=1))
    _write()
    del m
    gc.collect(); torch.cuda.empty_cache()
    assert torch.isfinite(got).all() and torch.isfinite(hid).all()
    assert rep["visual_tokens_rel"]["hip"] <= 1.25 * rep["visual_tokens_rel"]["bf16_eager"] + 2e-3
    assert rep["final_hidden_rel"]["hip"] <= 1.25 * rep["final_hidden_rel"]["bf16_eager"] + 1e-3 and rep["final_hidden_rel"]["hip"] < 6e-2
    for h, e in zip(rep["decode_hip_vs_fp32_oracle"], rep["decode_bf16_eager_vs_fp32_oracle"]):
        assert h <= 1.5 * e + 1e-3 and h < 8e-2, (h, e)


def test_zz_no_silent_fulldepth_skip():
    """A full-depth case that skipped for lack of host memory must not leave the suite green unnoticed: the reason is in
    gpurun_out/fulldepth_r5_report.json ("skipped") and this test FAILS unless HVLM_ALLOW_FULLDEPTH_SKIP=1 says the box is known to be small."""
    _write()
    if SKIPPED and os.environ.get("HVLM_ALLOW_FULLDEPTH_SKIP") != "1":
        pytest.fail("full-depth parity cases skipped (set HVLM_ALLOW_FULLDEPTH_SKIP=1 to accept): " + json.dumps(SKIPPED))
