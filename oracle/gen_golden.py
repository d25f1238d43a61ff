#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference) on CPU through oracle/shim.py.

Run only in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
The fixtures hold inputs that are not re-derivable from a seed plus expected outputs (data only);
weights/batches come from oracle/synth.py (seeded numpy), so they are regenerated, not stored.

Fixtures:
  g1_pool.npz        VisualToTokenHelper.compress_tokens('temporal_spatial_pool')
  g2_splice.npz      prepare_inputs_labels_for_multimodal (train branch, 4 / 0 hand tokens, padded tail)
  g3_forward.npz     tiny HandsOnVLMForCausalLM forward+backward (CLIP -> projector -> pool -> splice ->
                     Llama -> lm_head/CE + CVAE): losses, logits/hidden slices, a few gradients
  g4_ops.npz         transformers LlamaDecoderLayer (dh=128) / CLIPEncoderLayer (dh=64) / CLIP tower
  g5_cvae.npz        TrajCVAE.forward / .inference at token_dim 2048 with fixed noise
  g6_sample.npz      HandsOnVLMForCausalLM.sample() trace with forced tokens (sequences, pred_hands)
  g7_traj_eval.npz   evaluate_traj_stochastic known answer (pins the ADE/FDE label swap)
  g3_forward_dh128.npz / g3_bf16_dh128.npz / g6_sample_dh128.npz   (`g23_dh128`, `g3bf16_dh128`)
                     the same end-to-end fixtures on synth.MID: D = 256, H = 2 -> dh = 128, the production head size
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shim  # noqa: E402

shim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrs):
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB", {k: v.shape for k, v in arrs.items()})


def load_clip(clip, cfg):
    """Loads the synthetic CLIP weights (reference / transformers-4.31 key names, `vision_model.*`) into
    the installed CLIPVisionModel, whose keys lost that prefix in transformers 5.x."""
    sd = synth.to_torch(synth.clip_params(cfg))
    have = set(clip.state_dict().keys())
    if not any(k.startswith("vision_model.") for k in have):
        sd = {k[len("vision_model."):]: v for k, v in sd.items()}
    missing, unexpected = clip.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k for k in missing), missing


# ------------------------------------------------------------------------------------------------
def g1_pool():
    from hoi_forecast.model.visual_to_tokens import VisualToTokenHelper
    r = np.random.RandomState(101)
    tokens = torch.from_numpy(r.standard_normal((1, 100, 256, 8)).astype(np.float32))
    h = VisualToTokenHelper(None, None, "origin", "temporal_spatial_pool", 8, 8)
    h.b, h.t = 1, 100
    out, mask = h.compress_tokens(tokens, torch.ones(1, 100, 256, dtype=torch.bool))
    assert out.shape == (1, 356, 8) and bool(mask.all())
    # the input is stored as fp16; the expected output is computed from the fp16-rounded input
    t16 = tokens.half().float()
    out16, _ = h.compress_tokens(t16, torch.ones(1, 100, 256, dtype=torch.bool))
    save("g1_pool.npz", tokens=tokens.half(), out=out16)


def build_reference_model(cfg):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from handsonvlm.model.language_model.handsonvlm import HandsOnVLMConfig, HandsOnVLMForCausalLM
    from llava.model.multimodal_encoder.clip_encoder import CLIPVisionTower

    hcfg = HandsOnVLMConfig(hidden_size=cfg["D"], intermediate_size=cfg["FFN"], num_hidden_layers=cfg["L"],
                            num_attention_heads=cfg["H"], vocab_size=cfg["vocab"], rms_norm_eps=1e-6)
    hcfg._attn_implementation = "eager"
    model = HandsOnVLMForCausalLM(hcfg, traj_decoder_name="CVAE")
    ccfg = CLIPVisionConfig(hidden_size=cfg["clip_d"], intermediate_size=cfg["clip_mlp"],
                            num_hidden_layers=cfg["clip_layers"], num_attention_heads=cfg["clip_heads"],
                            image_size=cfg["image"], patch_size=cfg["patch"], hidden_act="quick_gelu")
    ccfg._attn_implementation = "eager"
    clip = CLIPVisionModel(ccfg)
    load_clip(clip, cfg)
    clip.requires_grad_(False)
    tower = CLIPVisionTower.__new__(CLIPVisionTower)
    nn.Module.__init__(tower)
    tower.is_loaded = True
    tower.vision_tower = clip
    tower.select_layer = -2
    tower.select_feature = "patch"
    # Stored as a one-element list (the reference's own fsdp code path, llava/model/llava_arch.py:51-54;
    # get_vision_tower() unwraps it): this keeps the CLIP modules out of LlamaModel's module tree.
    # With transformers 5.x a CLIP registered under LlamaModel gets its hidden-state capture hooks
    # installed twice after the first LlamaModel.forward, so hidden_states[-2] silently selects the
    # wrong layer from the second forward on (a version-drift artefact, not reference behaviour).
    model.model.vision_tower = [tower]
    model.model.mm_projector = nn.Linear(cfg["clip_d"], cfg["D"])
    params = synth.to_torch(synth.llama_params(cfg))
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    leftover = [k for k in missing if not any(s in k for s in (
        "vision_tower", "time_embedding", "hand_traj_positional_embedding", "bbox_to_feature", "feat_fusion",
        "downproject", "rotary_emb"))]
    assert not leftover, leftover
    c = model.config
    c.mm_hidden_size = cfg["clip_d"]
    c.fuse_input_mode = "origin"
    c.video_compress_mode = "temporal_spatial_pool"
    c.lambda_traj, c.lambda_traj_kl, c.hoi_lambda = 1.0, 1e-3, 1.0
    c.use_cache = False
    return model


def g2_g3(cfg, tag=""):
    """tag "": the TINY family (dh = 64), also writes g2_splice; tag "_dh128": synth.MID (dh = 128), g3 / g6 only."""
    model = build_reference_model(cfg)
    model.train()
    B, T = 2, 24
    ids, labels, mask = synth.make_text_batch(B, T, seed=3, n_hand=(4, 0), pad_tail=(0, 5))
    images = synth.make_images(B, cfg["frames"], 10, cfg["image"], seed=11)
    fh, fv = synth.make_hands(B, seed=5)
    contact = torch.zeros(B, 2)
    glv = torch.ones(B)

    # ---- G2: the splice alone (train branch) ---------------------------------------------------
    model.B = B
    with torch.no_grad():
        _, am, _, emb, lab = model.prepare_inputs_labels_for_multimodal(
            ids, mask, None, labels, images, future_hands=fh[:, :, 1:, :], future_valid=fv.clone(), is_evaluate=False)
    if not tag:
        save("g2_splice.npz", inputs_embeds=emb, labels=lab, attention_mask=am.to(torch.int64))

    # ---- G3: full forward + backward -------------------------------------------------------------
    torch.manual_seed(123)
    eps_noise = torch.randn(B * 8, 256)          # what reparameterize's randn_like will draw next
    torch.manual_seed(123)
    out = model(input_ids=ids, attention_mask=mask, labels=labels, image=images, future_hands=fh.clone(),
                contact_point=contact, future_valid=fv.clone(), gt_label_valid=glv, output_hidden_states=True,
                return_dict=True)
    out.loss.backward()
    sd = dict(model.named_parameters())
    # recompute the loss components the same way the reference does, for the record
    hidden = out.hidden_states[-1]
    save(f"g3_forward{tag}.npz",
         loss=out.loss, eps_noise=eps_noise,
         logits_slice=out.logits[:, ::7, ::251], logits_last=out.logits[:, -1, :],
         hidden=hidden,
         g_proj_w=sd["model.mm_projector.weight"].grad, g_proj_b=sd["model.mm_projector.bias"].grad,
         g_q0=sd["model.layers.0.self_attn.q_proj.weight"].grad,
         g_down1=sd["model.layers.1.mlp.down_proj.weight"].grad,
         g_ln0=sd["model.layers.0.input_layernorm.weight"].grad,
         g_norm=sd["model.norm.weight"].grad,
         g_embed_hand=sd["model.embed_tokens.weight"].grad[32100],
         g_embed_rows=sd["model.embed_tokens.weight"].grad[ids[0, :5].clamp(min=0)],
         g_lm_head_rows=sd["lm_head.weight"].grad[::1601],
         g_cvae_enc=sd["hand_traj_decoder.hand_traj_decoder.cvae.enc_MLP.0.weight"].grad[::8, ::4],
         g_cvae_dec2=sd["hand_traj_decoder.hand_traj_decoder.cvae.dec_MLP.2.weight"].grad)

    # ---- G6: sample() with forced tokens ----------------------------------------------------------
    from transformers import LogitsProcessor, LogitsProcessorList, StoppingCriteriaList, MaxLengthCriteria
    model.eval()
    forced = [512, 32100, 32100, 77, 32100, 32100, 2]
    prompt = ids[:1, :12].clone()

    class Force(LogitsProcessor):
        def __init__(self, n0): self.n0 = n0
        def __call__(self, input_ids, scores):
            tok = forced[input_ids.shape[1] - self.n0]
            s = torch.full_like(scores, -1e9); s[:, tok] = 0.0
            return s

    zs = []
    real_randn = torch.randn

    def rec_randn(*a, **k):
        z = real_randn(*a, **k); zs.append(z.clone()); return z

    torch.manual_seed(7)
    torch.randn = rec_randn
    try:
        with torch.no_grad():
            so = model.sample(prompt, logits_processor=LogitsProcessorList([Force(prompt.shape[1])]),
                              stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(prompt.shape[1] + len(forced))]),
                              pad_token_id=0, eos_token_id=2, output_hidden_states=True, return_dict_in_generate=True,
                              image=images[:1], use_cache=False,
                              attention_mask=torch.ones_like(prompt, dtype=torch.bool))
    finally:
        torch.randn = real_randn
    save(f"g6_sample{tag}.npz", prompt=prompt, forced=np.array(forced), sequences=so.sequences, pred_hands=so.pred_hands,
         z=torch.stack(zs, 0))


def g4_ops():
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding
    cfg = dict(D=256, H=2, L=1, FFN=512, vocab=8, clip_d=64)
    lc = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2,
                     num_key_value_heads=2, vocab_size=8, rms_norm_eps=1e-6)
    lc._attn_implementation = "eager"
    layer = LlamaDecoderLayer(lc, 0)
    p = synth.llama_params(dict(cfg, vocab=8), seed=21)
    sd = {k[len("model.layers.0."):]: torch.from_numpy(v) for k, v in p.items() if k.startswith("model.layers.0.")}
    layer.load_state_dict(sd)
    r = np.random.RandomState(22)
    x = torch.from_numpy(r.standard_normal((2, 40, 256)).astype(np.float32)).requires_grad_(True)
    rot = LlamaRotaryEmbedding(lc)
    pos = torch.arange(40)[None]
    cos, sin = rot(x, pos)
    S = 40
    causal = torch.triu(torch.full((S, S), float("-inf")), 1)[None, None]
    y = layer(x, attention_mask=causal, position_ids=pos, position_embeddings=(cos, sin))
    y = y[0] if isinstance(y, tuple) else y
    dy = torch.from_numpy(r.standard_normal((2, 40, 256)).astype(np.float32))
    y.backward(dy)
    g = dict(layer.named_parameters())
    # CLIP tower (3 layers, d=64, 1 head -> dh=64) on 2 images
    ccfg = synth.TINY
    cc = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=1,
                          image_size=224, patch_size=14, hidden_act="quick_gelu")
    cc._attn_implementation = "eager"
    clip = CLIPVisionModel(cc)
    load_clip(clip, ccfg)
    img = synth.make_images(1, 2, 2, 224, seed=31)[0]
    with torch.no_grad():
        hs = clip(img, output_hidden_states=True).hidden_states
    save("g4_ops.npz", x=x.detach(), dy=dy, y=y.detach(), dx=x.grad, g_q=g["self_attn.q_proj.weight"].grad,
         g_gate=g["mlp.gate_proj.weight"].grad, g_ln2=g["post_attention_layernorm.weight"].grad,
         cos=cos[0], sin=sin[0], clip_feat=hs[-2][:, 1:], clip_h0=hs[0], clip_h1=hs[1])


def g5_cvae():
    from handsonvlm.model.language_model.traj_decoder import CVAETrajDecoder
    dec = CVAETrajDecoder(token_dim=2048)
    p = synth.to_torch(synth.cvae_params(2048, seed=41))
    missing, unexpected = dec.load_state_dict({k[len("hand_traj_decoder."):]: v for k, v in p.items()}, strict=True)
    r = np.random.RandomState(42)
    B = 3
    emb = torch.from_numpy(r.standard_normal((B, 2, 4, 2048)).astype(np.float32) * 0.5).half().float()  # stored as fp16
    fh = torch.from_numpy(r.uniform(0, 1, (B, 2, 4, 2)).astype(np.float32))
    fv = torch.tensor([[1., 1.], [1., 0.], [0., 0.]])
    torch.manual_seed(5)
    eps_noise = torch.randn(B * 8, 256)
    torch.manual_seed(5)
    loss = dec(pred_hand_embeddings=emb, future_hands=fh, future_valid=fv, lambda_traj=1.0, lambda_traj_kl=1e-3)
    torch.manual_seed(6)
    z = 2.0 * torch.randn(B * 2 * 1, 256)
    torch.manual_seed(6)
    with torch.no_grad():
        inf = dec.inference(pred_hand_embeddings=emb[:, :, :1])
    save("g5_cvae.npz", emb=emb.half(), fh=fh, fv=fv, eps_noise=eps_noise, z=z, traj_loss=loss["traj_loss"],
         kl_loss=loss["traj_kl_loss"], total=loss["total_loss"], inference=inf)


def g7_traj_eval():
    from hoi_forecast.evaluation.traj_eval import evaluate_traj_stochastic
    r = np.random.RandomState(0)
    preds = r.rand(5, 1, 2, 4, 2).astype(np.float32)
    gts = (r.rand(5, 2, 4, 2).astype(np.float32) * 1.2 - 0.1)
    valids = (r.rand(5, 2) > 0.3).astype(np.float32)
    valids[0] = 1
    a, f, w = evaluate_traj_stochastic(preds, gts, valids)
    preds3 = r.rand(6, 3, 2, 4, 2).astype(np.float32)
    gts3 = r.rand(6, 2, 4, 2).astype(np.float32)
    val3 = np.ones((6, 2), dtype=np.float32); val3[2, 1] = 0
    a3, f3, w3 = evaluate_traj_stochastic(preds3, gts3, val3)
    save("g7_traj_eval.npz", preds=preds, gts=gts, valids=valids, ret=np.array([a, f, w], dtype=np.float64),
         preds3=preds3, gts3=gts3, valids3=val3, ret3=np.array([a3, f3, w3], dtype=np.float64))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g23", "g4", "g5", "g7"]
    if "g1" in which: g1_pool()
    if "g7" in which: g7_traj_eval()
    if "g5" in which: g5_cvae()
    if "g4" in which: g4_ops()
    if "g23" in which: g2_g3(synth.TINY)
    if "g23_dh128" in which: g2_g3(synth.MID, "_dh128")


def g3_bf16(cfg, tag=""):
    """The bf16-EAGER envelope of G3: the same reference model and batch as g2_g3, but run the way the reference trains
    (`--bf16 True`, scripts/finetune.sh:15): every LLM / projector / head module in bfloat16, the CLIP tower in float16
    (handsonvlm/train/train.py:316), PyTorch eager kernels with their own fp32 accumulation.  Its distance from the fp32
    golden (g3_forward.npz) is the rounding noise a bf16 implementation of this path is EXPECTED to have; the GPU tests
    bound the HIP path's distance from the fp32 golden by a small multiple of it (tests/test_gpu_model.py)."""
    model = build_reference_model(cfg)
    model.train()
    model.to(torch.bfloat16)
    model.model.vision_tower[0].to(torch.float16)
    B, T = 2, 24
    ids, labels, mask = synth.make_text_batch(B, T, seed=3, n_hand=(4, 0), pad_tail=(0, 5))
    images = synth.make_images(B, cfg["frames"], 10, cfg["image"], seed=11).to(torch.bfloat16)
    fh, fv = synth.make_hands(B, seed=5)
    # the SAME CVAE noise as G3 (randn_like on a bf16 tensor would draw a different stream): the fp32 draw, cast
    torch.manual_seed(123)
    eps_noise = torch.randn(B * 8, 256)
    real = torch.randn_like
    torch.randn_like = lambda t, **k: eps_noise.to(t.dtype) if tuple(t.shape) == tuple(eps_noise.shape) else real(t, **k)
    try:
        out = model(input_ids=ids, attention_mask=mask, labels=labels, image=images, future_hands=fh.clone().to(torch.bfloat16),
                    contact_point=torch.zeros(B, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(B),
                    output_hidden_states=True, return_dict=True)
    finally:
        torch.randn_like = real
    out.loss.backward()
    sd = dict(model.named_parameters())
    f = lambda t: t.detach().float()  # noqa: E731
    save(f"g3_bf16{tag}.npz", loss=f(out.loss), logits_slice=f(out.logits[:, ::7, ::251]), hidden=f(out.hidden_states[-1]),
         g_proj_w=f(sd["model.mm_projector.weight"].grad), g_proj_b=f(sd["model.mm_projector.bias"].grad),
         g_q0=f(sd["model.layers.0.self_attn.q_proj.weight"].grad),
         g_down1=f(sd["model.layers.1.mlp.down_proj.weight"].grad),
         g_ln0=f(sd["model.layers.0.input_layernorm.weight"].grad), g_norm=f(sd["model.norm.weight"].grad),
         g_embed_hand=f(sd["model.embed_tokens.weight"].grad[32100]),
         g_embed_rows=f(sd["model.embed_tokens.weight"].grad[ids[0, :5].clamp(min=0)]),
         g_lm_head_rows=f(sd["lm_head.weight"].grad[::1601]))


if __name__ == "__main__" and "g3bf16" in sys.argv[1:]:
    g3_bf16(synth.TINY)
if __name__ == "__main__" and "g3bf16_dh128" in sys.argv[1:]:
    g3_bf16(synth.MID, "_dh128")
