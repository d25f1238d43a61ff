"""GPU parity of the full hot path (HIP kernels behind HandsOnVLMForCausalLM) against
  (1) the golden vectors produced by the REFERENCE on CPU (tests/golden/g3_forward.npz, g6_sample.npz) and
  (2) the CPU oracle (oracle/ref_cpu.py) on the same seeded inputs.
The HIP path computes in bf16 with fp32 accumulation; the golden/oracle values are fp32, so the bounds
below are bf16 bounds (stated per quantity).  Run with `pytest -m gpu` on the MI355X box."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle import synth  # noqa: E402

T = torch.from_numpy
REPORT = {}


def rel(a, b):
    a = a.detach().float().cpu(); b = torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


FAMILIES = {"": synth.TINY, "_dh128": synth.MID}     # fixture tag -> config: dh = 64, and the production head size dh = 128


def tiny_model(training=True, family=None, **cfg_kw):
    from hvlm.config import CLIPVisionConfig, HandsOnVLMConfig
    from hvlm.model import HandsOnVLMForCausalLM
    c = family or synth.TINY
    cfg = HandsOnVLMConfig(hidden_size=c["D"], intermediate_size=c["FFN"], num_hidden_layers=c["L"],
                           num_attention_heads=c["H"], mm_hidden_size=c["clip_d"],
                           vision=CLIPVisionConfig(hidden_size=c["clip_d"], intermediate_size=c["clip_mlp"],
                                                   num_hidden_layers=c["clip_layers"], num_attention_heads=c["clip_heads"]),
                           **cfg_kw)
    m = HandsOnVLMForCausalLM(cfg, device="cuda:0", training=training)
    sd = synth.to_torch(synth.llama_params(c))
    sd.update({"model.vision_tower.vision_tower." + k: v for k, v in synth.to_torch(synth.clip_params(c)).items()})
    m.load_state_dict(sd)
    return m, cfg


def tiny_batch():
    c = synth.TINY
    ids, labels, mask = synth.make_text_batch(2, 24, seed=3, n_hand=(4, 0), pad_tail=(0, 5))
    images = synth.make_images(2, c["frames"], 10, c["image"], seed=11)
    fh, fv = synth.make_hands(2, seed=5)
    return ids, labels, mask, images, fh, fv


@pytest.mark.parametrize("tag", list(FAMILIES))
@pytest.mark.parametrize("project_after_pool", [True, False])
def test_forward_backward_vs_reference_golden(golden, project_after_pool, tag):
    """tag "_dh128": the same end-to-end check on synth.MID (D = 256, H = 2 -> dh = 128): the causal dh-128 attention forward /
    backward kernels of the 7B / 13B decoder inside a REFERENCE-pinned end-to-end run, with the same envelope assertion."""
    g3 = golden(f"g3_forward{tag}.npz")
    m, cfg = tiny_model(project_after_pool=project_after_pool, family=FAMILIES[tag])
    ids, labels, mask, images, fh, fv = tiny_batch()
    out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh.clone(),
            contact_point=torch.zeros(2, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(2),
            output_hidden_states=True, cvae_eps=T(g3["eps_noise"]).cuda(), prompt=["ignored"], feat=None)
    S = out.logits.shape[1]
    valid = torch.ones(2, S, dtype=torch.bool); valid[1, S - 5:] = False
    hid = out.hidden_states[-1].float().cpu()
    r = {"loss": float(out.loss), "loss_ref": float(g3["loss"]),
         "hidden_rel": rel(hid[valid], g3["hidden"][valid.numpy()]),
         "logits_rel": rel(out.logits.float().cpu()[:, ::7, ::251][valid[:, ::7]], g3["logits_slice"][valid[:, ::7].numpy()])}
    out.loss.backward()
    named = {n: g for n, w, g, wt in m.trainable_tensors()}
    D = cfg.hidden_size
    r["g_proj_w"] = rel(named["proj_w"], g3["g_proj_w"])
    r["g_proj_b"] = rel(named["proj_b"], g3["g_proj_b"])
    r["g_q0"] = rel(named["layers.0.qkv"][:D], g3["g_q0"])
    r["g_down1"] = rel(named["layers.1.down"], g3["g_down1"])
    r["g_ln0"] = rel(named["layers.0.ln1"], g3["g_ln0"])
    r["g_norm"] = rel(named["norm"], g3["g_norm"])
    r["g_embed_hand"] = rel(named["embed"][32100], g3["g_embed_hand"])
    r["g_embed_rows"] = rel(named["embed"][ids[0, :5].clamp(min=0).cuda()], g3["g_embed_rows"])
    r["g_lm_head_rows"] = rel(named["lm_head"][:32101][::1601], g3["g_lm_head_rows"])
    hp = dict(m.hand_traj_decoder.named_parameters())
    r["g_cvae_enc"] = rel(hp["hand_traj_decoder.cvae.enc_MLP.0.weight"].grad[::8, ::4], g3["g_cvae_enc"])
    r["g_cvae_dec2"] = rel(hp["hand_traj_decoder.cvae.dec_MLP.2.weight"].grad, g3["g_cvae_dec2"])
    REPORT[f"g3{tag}_pap{int(project_after_pool)}"] = r
    print(json.dumps(r, indent=1))
    # Parity statement (north_star: <= 1e-3 rel in bf16 — reachable per op, tests/test_gpu_ops.py; end to end the yardstick is
    # what bf16 itself costs on this path): g3_bf16.npz is the REFERENCE run bf16-eager (LLM bf16, tower fp16, as it trains;
    # oracle/gen_golden.py::g3_bf16) on the same inputs; its distance from the fp32 golden is the envelope.  The HIP path
    # must sit within 1.25x of that envelope per tensor, and under absolute caps at 1.5x the round-1 measurements.
    env = golden(f"g3_bf16{tag}.npz")
    envelope = {"hidden_rel": rel(T(env["hidden"])[valid], g3["hidden"][valid.numpy()]),
                "logits_rel": rel(T(env["logits_slice"])[valid[:, ::7]], g3["logits_slice"][valid[:, ::7].numpy()])}
    for k in ("g_proj_w", "g_proj_b", "g_q0", "g_down1", "g_ln0", "g_norm", "g_embed_hand", "g_embed_rows", "g_lm_head_rows"):
        envelope[k] = rel(T(env[k]), g3[k])
    r["bf16_eager_reference_envelope"] = envelope
    REPORT[f"g3{tag}_pap{int(project_after_pool)}"] = r
    assert abs(r["loss"] - r["loss_ref"]) / r["loss_ref"] < 2e-3           # bf16-eager reference: 6.3e-3
    for k, e in envelope.items():
        assert r[k] <= 1.25 * e, (k, r[k], e)
    assert r["hidden_rel"] < 1.3e-2 and r["logits_rel"] < 1.5e-2
    for k in ("g_proj_w", "g_proj_b", "g_q0", "g_down1", "g_ln0", "g_norm", "g_embed_hand", "g_embed_rows",
              "g_lm_head_rows", "g_cvae_enc", "g_cvae_dec2"):
        assert r[k] < 3e-2, (k, r[k])


@pytest.mark.parametrize("pack", [False, True])
def test_lm_head_labelled_rows_equals_full_logits_and_reference_gradients(golden, pack):
    """config.lm_head_labelled_rows (bench.py's default): lm_head / cross-entropy / their backward only on the rows that carry a
    target.  Same loss as the full-logits pass (fp32 sums over a different row count: <= 1e-6 rel), every gradient within the
    bf16 noise of the full pass, and the gradients still inside the REFERENCE golden's bounds (handsonvlm.py:132-145 computes all
    rows and lets CrossEntropyLoss drop the ignored ones).  Padded and packed layouts."""
    g3 = golden("g3_forward.npz")
    ids, labels, mask, images, fh, fv = tiny_batch()
    res = {}
    for flag in (False, True):
        m, cfg = tiny_model(lm_head_labelled_rows=flag, pack_sequences=pack)
        out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh.clone(),
                contact_point=torch.zeros(2, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(2),
                cvae_eps=T(g3["eps_noise"]).cuda())
        if flag:
            with pytest.raises(RuntimeError, match="lm_head_labelled_rows"):
                out.logits
            assert out.logits_rows.shape[0] == out.rows.numel()
        else:
            assert out.logits is not None
        loss = float(out.loss)
        out.loss.backward()
        g = {n: gr.float().clone() for n, _, gr, _ in m.trainable_tensors()}
        g.update({"head." + n: p.grad.float().clone() for n, p in m.hand_traj_decoder.named_parameters()})
        res[flag] = (loss, g, cfg)
    (l0, g0, cfg), (l1, g1, _) = res[False], res[True]
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    for n in g0:
        r = float((g0[n] - g1[n]).norm() / (g0[n].norm() + 1e-30))
        assert r < 4e-3, (n, r)
    assert abs(l1 - float(g3["loss"])) / float(g3["loss"]) < 2e-3
    assert rel(g1["lm_head"][:32101][::1601], g3["g_lm_head_rows"]) < 3e-2
    assert rel(g1["layers.1.down"], g3["g_down1"]) < 3e-2 and rel(g1["norm"], g3["g_norm"]) < 3e-2
    assert rel(g1["embed"][32100], g3["g_embed_hand"]) < 3e-2


def test_mixed_text_only_batch_backward_vs_oracle():
    """A text-only sample (no <image> token; handsonvlm.py:234-244) next to an image sample: its visual-token gradient
    rows are zero, so the projector gradient is that of the image sample alone — checked against the oracle's autograd on
    the same batch, with the allocator's free memory poisoned with NaNs first."""
    m, cfg = tiny_model()
    c = synth.TINY
    ids, labels, mask = synth.make_text_batch(2, 24, seed=3, n_hand=(4, 4))
    ids[1][ids[1] == -200] = 7
    images = synth.make_images(2, c["frames"], 10, c["image"], seed=11)
    fh, fv = synth.make_hands(2, seed=5)
    eps = torch.randn(16, 256, generator=torch.Generator().manual_seed(2))
    poison = torch.full((64 << 20,), float("nan"), dtype=torch.bfloat16, device="cuda")
    del poison                                                         # back to the caching allocator, bits intact
    out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh.clone(),
            contact_point=torch.zeros(2, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(2), cvae_eps=eps.cuda())
    out.loss.backward()
    named = {n: g for n, w, g, wt in m.trainable_tensors()}
    p = {k: v.clone().requires_grad_(True) for k, v in synth.to_torch(synth.llama_params(c)).items()}
    cp = synth.to_torch(synth.clip_params(c))
    vis = R.visual_tokens(images, cp, c["clip_layers"], c["clip_heads"], p["model.mm_projector.weight"], p["model.mm_projector.bias"])
    ref = R.forward_loss(p, dict(n_layers=c["L"], n_heads=c["H"], cond_dtype=torch.bfloat16), ids, labels, mask, vis, fh, fv, eps)
    ref["loss"].backward()
    assert abs(float(out.loss) - float(ref["loss"])) / float(ref["loss"]) < 1e-2
    for ours, theirs in (("proj_w", "model.mm_projector.weight"), ("proj_b", "model.mm_projector.bias")):
        assert torch.isfinite(named[ours].float()).all()
        assert rel(named[ours], p[theirs].grad) < 3e-2, (ours, rel(named[ours], p[theirs].grad))


@pytest.mark.parametrize("tag", list(FAMILIES))
@pytest.mark.parametrize("use_kv_cache", [True, False])
def test_sample_vs_reference_golden(golden, use_kv_cache, tag):
    """The reference's sample() re-runs the whole prefix per token (use_cache=False); both our cache-less loop and the
    KV-cached loop (prefill + single-token decode kernels) must reproduce its trace.  tag "_dh128": the reference's trace on
    synth.MID — sample() pinned at the production head size (dh = 128 decode attention / GEMV kernels)."""
    g = golden(f"g6_sample{tag}.npz")
    m, cfg = tiny_model(training=False, family=FAMILIES[tag])
    ids, labels, mask, images, fh, fv = tiny_batch()
    forced = [int(t) for t in g["forced"]]
    n0 = g["prompt"].shape[1]

    def force(input_ids, scores):
        s = torch.full_like(scores, -1e9)
        s[:, forced[input_ids.shape[1] - n0]] = 0.0
        return s

    z = [2.0 * T(g["z"][i]) for i in range(g["z"].shape[0])]        # z_scale = 2
    so = m.sample(T(g["prompt"]), logits_processor=[force], max_length=n0 + len(forced), pad_token_id=0, eos_token_id=2,
                  image=images[:1].to(torch.bfloat16), hand_noise=z, use_kv_cache=use_kv_cache)
    assert np.array_equal(so.sequences.cpu().numpy(), g["sequences"])
    assert so.pred_hands.shape == (1, 2, 4, 2)
    r = rel(so.pred_hands, g["pred_hands"])
    REPORT[f"g6{tag}_pred_hands_rel_kv{int(use_kv_cache)}"] = r
    print("pred_hands rel", r, so.pred_hands.flatten().tolist(), g["pred_hands"].flatten().tolist())
    assert r < 5e-2     # bf16 hidden state feeding an fp32 head
    # "eval traj ADE/FDE parity" (BASELINE.json metric): the metrics of the HIP trajectory vs those of the
    # reference's trajectory against the same ground truth, with the reference's metric code semantics.
    from hvlm.traj_eval import evaluate_traj_stochastic
    gt = fh[:1, :, 1:].numpy()
    valid = np.ones((1, 2), np.float32)
    ours = evaluate_traj_stochastic(so.pred_hands.cpu().numpy()[:, None], gt, valid)
    ref = R.evaluate_traj_stochastic(g["pred_hands"][:, None], gt, valid)
    REPORT[f"g6{tag}_ade_fde_wde_kv{int(use_kv_cache)}"] = dict(hip=[float(x) for x in ours], reference=[float(x) for x in ref])
    np.testing.assert_allclose(ours, ref, rtol=1e-2)


def test_cvae_head_on_gpu_within_ulps_of_reference_given_identical_inputs(golden):
    """north_star asks for trajectory coordinates bit-exact in fp32.  That holds on a CPU (tests/test_abi_and_host.py::
    test_cvae_head_matches_reference_golden_cpu: array_equal with the reference's output).  On the GPU the head's three
    nn.Linear run on hipBLASLt, whose fp32 summation order over K = 2304 / 512 differs from the CPU GEMM's, so bit equality
    with a CPU-generated golden is unreachable through the vendor library; what is asserted is an ULP bound: every decoded
    coordinate within 32 ulp of the LARGEST coordinate's magnitude (2^-18 relative to max |x|), the loss sums within 2e-6."""
    from hvlm.traj_decoder import CVAETrajDecoder
    g = golden("g5_cvae.npz")
    dec = CVAETrajDecoder(token_dim=2048)
    p = synth.to_torch(synth.cvae_params(2048, seed=41))
    dec.load_state_dict({k[len("hand_traj_decoder."):]: v for k, v in p.items()})
    dec = dec.cuda()
    emb = T(g["emb"]).float().cuda()
    loss = dec(emb, T(g["fh"]).cuda(), T(g["fv"]).cuda(), 1.0, 1e-3, eps=T(g["eps_noise"]).cuda())
    inf = dec.inference(emb[:, :, :1], T(g["z"]).cuda()).detach().cpu().numpy()
    np.testing.assert_allclose(loss["traj_loss"].item(), g["traj_loss"], rtol=2e-6)
    np.testing.assert_allclose(loss["traj_kl_loss"].item(), g["kl_loss"], rtol=2e-6)
    err = np.abs(inf - g["inference"]).max()
    ulp32 = 2.0 ** -18 * np.abs(g["inference"]).max()
    REPORT["g5_cvae_gpu_max_err_over_ulp_of_max"] = float(err / (2.0 ** -23 * np.abs(g["inference"]).max()))
    assert err <= ulp32, (err, ulp32)


def test_dedup_frames_identical():
    m, cfg = tiny_model(training=False)
    _, _, _, images, _, _ = tiny_batch()
    img = images.to(torch.bfloat16).cuda()
    a = m.visual_tokens(img)
    m.config.dedup_frames = True
    b = m.visual_tokens(img)
    assert torch.equal(a, b)




def test_336px_generalisation_vs_oracle():
    """24x24 patches per frame (336 px): CLIP on 577 tokens, 100 + 4*12*12 = 676 visual tokens.  The reference cannot
    run this geometry (it asserts 224 px / 256 patches), so the checker is the oracle's restatement of the same
    arithmetic on a 24x24 grid; the HIP path must agree with it like at 224 px."""
    from hvlm.config import CLIPVisionConfig, HandsOnVLMConfig
    from hvlm.model import HandsOnVLMForCausalLM
    c = dict(synth.TINY, image=336)
    cfg = HandsOnVLMConfig(hidden_size=c["D"], intermediate_size=c["FFN"], num_hidden_layers=c["L"],
                           num_attention_heads=c["H"], mm_hidden_size=c["clip_d"],
                           vision=CLIPVisionConfig(hidden_size=c["clip_d"], intermediate_size=c["clip_mlp"],
                                                   num_hidden_layers=c["clip_layers"], num_attention_heads=c["clip_heads"],
                                                   image_size=336))
    m = HandsOnVLMForCausalLM(cfg, device="cuda:0", training=True)
    p = synth.to_torch(synth.llama_params(c))
    cp = synth.to_torch(synth.clip_params(c))
    sd = dict(p)
    sd.update({"model.vision_tower.vision_tower." + k: v for k, v in cp.items()})
    m.load_state_dict(sd)
    ids, labels, mask = synth.make_text_batch(1, 24, seed=3, n_hand=(4,))
    images = synth.make_images(1, c["frames"], 2, 336, seed=11)
    fh, fv = synth.make_hands(1, seed=5)
    eps = torch.randn(8, 256, generator=torch.Generator().manual_seed(0))
    out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh.clone(),
            contact_point=torch.zeros(1, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(1), cvae_eps=eps.cuda())
    assert out.logits.shape[1] == 24 + 676 - 1
    out.loss.backward()
    vis = R.visual_tokens(images, cp, c["clip_layers"], c["clip_heads"], p["model.mm_projector.weight"],
                          p["model.mm_projector.bias"])
    assert vis.shape[1] == 676
    ref = R.forward_loss(p, dict(n_layers=c["L"], n_heads=c["H"], cond_dtype=torch.bfloat16), ids, labels, mask, vis, fh, fv, eps)
    got, want = float(out.loss.detach()), float(ref["loss"])
    REPORT["336px_loss"] = dict(hip=got, oracle=want)
    assert abs(got - want) / abs(want) < 1e-2, (got, want)


def test_zz_write_report():
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_model_report.json", "w") as f:
        json.dump(REPORT, f, indent=1)


def test_graph_replayed_decode_bit_identical_to_eager(monkeypatch):
    """The decode step replayed from ONE captured hipGraph (position on the device, hvlm_llama_decode_step_dev) against the
    eagerly launched step (hvlm_llama_decode_step): hidden states and the KV rows it writes bit-identical over several
    tokens, across a restart at an earlier position (next sample), with the cache's last row untouched by the capture."""
    from hvlm import _lib
    m, cfg = tiny_model(training=False)
    st, D = m.stack, cfg.hidden_size
    cache = st.new_cache(64)
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(20, D, device="cuda", generator=gen).to(torch.bfloat16)
    for c in cache:
        c.copy_(torch.randn(c.shape, device="cuda", generator=gen).to(torch.bfloat16))
    st.forward(x, 1, 20, save=False, cache=cache)
    snap = [c.clone() for c in cache]
    xs = [torch.randn(1, D, device="cuda", generator=gen).to(torch.bfloat16) for _ in range(4)]
    res = {}
    for mode in ("0", "1", "unfused"):
        # "unfused": RoPE and the combine of the key splits as separate launches (the fused attention kernel is the default)
        monkeypatch.setenv("HVLM_DECODE_GRAPH", "0" if mode == "unfused" else mode)
        _lib.configure("decode_unfused", 1 if mode == "unfused" else 0)
        for c, s0 in zip(cache, snap):
            c.copy_(s0)
        hs = [st.decode_step(x1, 20 + i, cache).clone() for i, x1 in enumerate(xs)]
        rows = [c[20:24].clone() for c in cache]
        hs.append(st.decode_step(xs[1], 20, cache).clone())          # a new sample restarts at an earlier position
        hs.append(st.decode_step(xs[2], 21, cache).clone())
        res[mode] = (hs, rows, [c[63].clone() for c in cache])
    _lib.configure("decode_unfused", 0)
    assert st._dec_graph is not None
    D3 = 2 * D                                                       # compare the K and V parts of the rows (q is scratch)
    for other in ("1", "unfused"):
        for a, b in zip(res["0"][0], res[other][0]):
            assert torch.equal(a, b) and torch.isfinite(a.float()).all(), other
        for a, b in zip(res["0"][1], res[other][1]):
            assert torch.equal(a[:, D:D + D3], b[:, D:D + D3]), other
    for a, s0 in zip(res["1"][2], snap):
        assert torch.equal(a, s0[63])
