"""Pins oracle/ref_cpu.py against outputs of the reference itself (tests/golden/, produced by
oracle/gen_golden.py running /root/reference on CPU).  CPU-only; no reference import here."""
import numpy as np
import torch

from oracle import ref_cpu as R
from oracle import synth

T = torch.from_numpy


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_g1_temporal_spatial_pool(golden):
    g = golden("g1_pool.npz")
    out = R.temporal_spatial_pool(T(g["tokens"]).float())
    assert out.shape == (1, 356, 8)
    close(out, g["out"], rtol=1e-6, atol=1e-6)


def test_g7_traj_eval_label_swap(golden):
    g = golden("g7_traj_eval.npz")
    ret = R.evaluate_traj_stochastic(g["preds"], g["gts"], g["valids"])
    close(np.array(ret), g["ret"], rtol=1e-6)
    # the values BASELINE.md quotes for this seed: reference returns (FDE, ADE, WDE)
    ade, fde, wde = R.evaluate_traj_stochastic(g["preds"], g["gts"], g["valids"], reference_labels=False)
    assert np.isclose(ade, ret[1]) and np.isclose(fde, ret[0])      # swapped names
    close(np.array(R.evaluate_traj_stochastic(g["preds3"], g["gts3"], g["valids3"])), g["ret3"], rtol=1e-6)


def test_g5_cvae(golden):
    g = golden("g5_cvae.npz")
    p = synth.to_torch(synth.cvae_params(2048, seed=41))
    emb = T(g["emb"]).float()
    traj, kl = R.traj_losses(p, "hand_traj_decoder.hand_traj_decoder.cvae.", emb, T(g["fh"]), T(g["fv"]),
                             T(g["eps_noise"]), 1.0, 1e-3, cond_dtype=torch.bfloat16)
    close(traj, g["traj_loss"], rtol=1e-5)
    close(kl, g["kl_loss"], rtol=1e-5)
    close(traj + kl, g["total"], rtol=1e-5)
    cond = emb[:, :, :1].reshape(-1, 2048)
    inf = R.cvae_inference(p, "hand_traj_decoder.hand_traj_decoder.cvae.", cond, T(g["z"]))
    close(inf.reshape(3, 2, 1, 2), g["inference"], rtol=1e-5, atol=1e-6)   # bit-level in fp32 given identical inputs


def test_g4_llama_layer_and_clip(golden):
    g = golden("g4_ops.npz")
    cfg = dict(D=256, H=2, L=1, FFN=512, vocab=8, clip_d=64)
    p = synth.to_torch(synth.llama_params(cfg, seed=21))
    for v in p.values():
        v.requires_grad_(True)
    x = T(g["x"]).requires_grad_(True)
    cos, sin = R.rope_tables(128, 40)
    close(torch.cat([cos, cos], -1), g["cos"], rtol=1e-5, atol=1e-6)
    close(torch.cat([sin, sin], -1), g["sin"], rtol=1e-5, atol=1e-6)
    y = R.llama_layer(x, p, "model.layers.0.", 2, cos, sin, 1e-6)
    close(y, g["y"], rtol=1e-4, atol=1e-5)
    y.backward(T(g["dy"]))
    close(x.grad, g["dx"], rtol=1e-3, atol=1e-5)
    close(p["model.layers.0.self_attn.q_proj.weight"].grad, g["g_q"], rtol=1e-3, atol=1e-5)
    close(p["model.layers.0.mlp.gate_proj.weight"].grad, g["g_gate"], rtol=1e-3, atol=1e-5)
    close(p["model.layers.0.post_attention_layernorm.weight"].grad, g["g_ln2"], rtol=1e-3, atol=1e-5)
    # CLIP tower
    cp = synth.to_torch(synth.clip_params(synth.TINY))
    img = synth.make_images(1, 2, 2, 224, seed=31)[0]
    pre = "vision_model."
    h0 = R.clip_embeddings(img, cp, pre)
    h0 = torch.nn.functional.layer_norm(h0, (64,), cp[pre + "pre_layrnorm.weight"], cp[pre + "pre_layrnorm.bias"], 1e-5)
    close(h0, g["clip_h0"], rtol=1e-4, atol=1e-5)
    close(R.clip_layer(h0, cp, pre + "encoder.layers.0.", 1), g["clip_h1"], rtol=1e-4, atol=1e-5)
    close(R.clip_features(img, cp, 3, 1), g["clip_feat"], rtol=1e-4, atol=2e-5)


FAMILIES = {"": synth.TINY, "_dh128": synth.MID}     # fixture-name tag -> config (dh = 64 / the production dh = 128)


def _tiny_inputs(cfg=synth.TINY):
    B, Tn = 2, 24
    ids, labels, mask = synth.make_text_batch(B, Tn, seed=3, n_hand=(4, 0), pad_tail=(0, 5))
    images = synth.make_images(B, cfg["frames"], 10, cfg["image"], seed=11)
    fh, fv = synth.make_hands(B, seed=5)
    return cfg, ids, labels, mask, images, fh, fv


import pytest  # noqa: E402


@pytest.mark.parametrize("tag", list(FAMILIES))
def test_g2_g3_full_forward_backward(golden, tag):
    g2, g3 = (golden("g2_splice.npz") if not tag else None), golden(f"g3_forward{tag}.npz")
    cfg, ids, labels, mask, images, fh, fv = _tiny_inputs(FAMILIES[tag])
    p = synth.to_torch(synth.llama_params(cfg))
    cp = synth.to_torch(synth.clip_params(cfg))
    for v in p.values():
        v.requires_grad_(True)
    vis = R.visual_tokens(images, cp, cfg["clip_layers"], cfg["clip_heads"], p["model.mm_projector.weight"],
                          p["model.mm_projector.bias"])
    assert vis.shape == (2, 356, cfg["D"])
    x, y, lens = R.splice_batch(ids, labels, mask, p["model.embed_tokens.weight"], vis, fh[:, :, 1:], False)
    if g2 is not None:
        close(x, g2["inputs_embeds"], rtol=1e-4, atol=2e-5)
        assert np.array_equal(y.numpy(), g2["labels"])
    out = R.forward_loss(p, dict(n_layers=cfg["L"], n_heads=cfg["H"], cond_dtype=torch.bfloat16), ids, labels, mask, vis,
                         fh, fv, T(g3["eps_noise"]))
    # Sample 1 ends in 5 pad tokens whose keys the reference masks (attention_mask False); under the
    # causal mask that only changes the pad QUERY rows themselves, which carry no loss.  Compare the
    # valid rows (all of sample 0, all but the last 5 of sample 1).
    S = out["hidden"].shape[1]
    valid = torch.ones(2, S, dtype=torch.bool); valid[1, S - 5:] = False
    close(out["hidden"][valid], g3["hidden"][valid.numpy()], rtol=2e-4, atol=2e-5)
    vs = valid[:, ::7]
    close(out["logits"][:, ::7, ::251][vs], g3["logits_slice"][vs.numpy()], rtol=2e-4, atol=2e-5)
    close(out["logits"][0, -1, :], g3["logits_last"][0], rtol=2e-4, atol=5e-5)
    close(out["loss"], g3["loss"], rtol=1e-5)
    out["loss"].backward()
    G = lambda k: p[k].grad  # noqa: E731
    close(G("model.mm_projector.weight"), g3["g_proj_w"], rtol=2e-3, atol=1e-6)
    close(G("model.mm_projector.bias"), g3["g_proj_b"], rtol=2e-3, atol=1e-6)
    close(G("model.layers.0.self_attn.q_proj.weight"), g3["g_q0"], rtol=2e-3, atol=1e-6)
    close(G("model.layers.1.mlp.down_proj.weight"), g3["g_down1"], rtol=2e-3, atol=1e-6)
    close(G("model.layers.0.input_layernorm.weight"), g3["g_ln0"], rtol=2e-3, atol=1e-6)
    close(G("model.norm.weight"), g3["g_norm"], rtol=2e-3, atol=1e-6)
    close(G("model.embed_tokens.weight")[32100], g3["g_embed_hand"], rtol=2e-3, atol=1e-6)
    close(G("model.embed_tokens.weight")[ids[0, :5].clamp(min=0)], g3["g_embed_rows"], rtol=2e-3, atol=1e-6)
    close(G("lm_head.weight")[::1601], g3["g_lm_head_rows"], rtol=2e-3, atol=1e-6)
    close(G("hand_traj_decoder.hand_traj_decoder.cvae.enc_MLP.0.weight")[::8, ::4], g3["g_cvae_enc"], rtol=2e-3, atol=1e-6)
    close(G("hand_traj_decoder.hand_traj_decoder.cvae.dec_MLP.2.weight"), g3["g_cvae_dec2"], rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("tag", list(FAMILIES))
def test_g6_sample_trace(golden, tag):
    g = golden(f"g6_sample{tag}.npz")
    cfg, ids, labels, mask, images, fh, fv = _tiny_inputs(FAMILIES[tag])
    p = synth.to_torch(synth.llama_params(cfg))
    cp = synth.to_torch(synth.clip_params(cfg))
    vis = R.visual_tokens(images[:1], cp, cfg["clip_layers"], cfg["clip_heads"], p["model.mm_projector.weight"],
                          p["model.mm_projector.bias"])
    seq = T(g["prompt"]).clone()
    hands = []
    zi = 0
    mc = dict(n_layers=cfg["L"], n_heads=cfg["H"])
    for tok in g["forced"]:
        ph = torch.stack(hands, dim=2) if hands else None
        logits, last_h = R.generation_step(p, mc, seq, vis, ph)
        if int(tok) == R.HAND_TOKEN_ID:
            hands.append(R.decode_hand(p, last_h, 2.0 * T(g["z"][zi])))   # z_scale = 2 (traj_decoder.py:88)
            zi += 1
        seq = torch.cat([seq, torch.tensor([[int(tok)]])], dim=1)
    assert np.array_equal(seq.numpy(), g["sequences"])
    close(torch.stack(hands, dim=2), g["pred_hands"], rtol=1e-4, atol=1e-5)
