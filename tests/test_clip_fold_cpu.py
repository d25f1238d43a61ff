"""Host logic of the LayerNorm fold of the frozen CLIP tower (hvlm/clip.py::_fold_prepare / _forward_layers_folded) on the ops
stand-in (tests/cpu_ops.py): the folded tower must compute what the explicit one computes — CLIPEncoderLayer.layer_norm1 ->
q|k|v, layer_norm2 -> fc1 (transformers modeling_clip, through llava/model/multimodal_encoder/clip_encoder.py:39-51) — and both
must match the fp32 oracle.  The kernels themselves are tested on the GPU (tests/test_gpu_parity_r2.py)."""
import os

import pytest
import torch

from oracle import ref_cpu as R
from oracle import synth


def rel(a, b):
    a = torch.as_tensor(a).float().cpu(); b = torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("select_layer", [-2, -1])
def test_folded_tower_equals_explicit_tower_and_oracle(monkeypatch, select_layer):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import clip as clip_mod
    from hvlm.config import CLIPVisionConfig
    c = dict(clip_d=256, clip_mlp=512, clip_layers=4, clip_heads=4, patch=14, image=28)
    pnp = synth.clip_params(c, seed=5)
    p = synth.to_torch(pnp)
    # LayerNorm affines far from (1, 0) and an outlier channel, so that a wrong fold cannot hide
    for k in list(p):
        if "layer_norm" in k and k.endswith("weight"):
            p[k] = p[k] * (1.0 + 0.5 * torch.sin(torch.arange(p[k].numel(), dtype=torch.float32)))
        if "layer_norm" in k and k.endswith("bias"):
            p[k] = p[k] + 0.3 * torch.cos(torch.arange(p[k].numel(), dtype=torch.float32))
    img = synth.make_images(1, 5, 5, 28, seed=9)[0]                          # [5, 3, 28, 28]
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = R.clip_features(img, p, c["clip_layers"], c["clip_heads"], select_layer=select_layer) if select_layer != -2 \
            else R.clip_features(img, p, c["clip_layers"], c["clip_heads"])
    cfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=4, image_size=28, patch_size=14)
    outs = {}
    for fold in (True, False):
        monkeypatch.setattr(clip_mod, "_FOLD_LN", fold)
        tower = clip_mod.CLIPVisionTower(cfg, "cpu", select_layer=select_layer, dtype=torch.float16)
        tower.load_state_dict(p)
        calls = {"ln": 0, "fold": 0}
        from hvlm import ops
        ln0, g0 = ops.layernorm_fwd_f16, ops.gemm_f16_nt_ln
        monkeypatch.setattr(ops, "layernorm_fwd_f16", lambda *a, **k: (calls.__setitem__("ln", calls["ln"] + 1), ln0(*a, **k))[1])
        monkeypatch.setattr(ops, "gemm_f16_nt_ln", lambda *a, **k: (calls.__setitem__("fold", calls["fold"] + 1), g0(*a, **k))[1])
        outs[fold] = tower(img.to(torch.bfloat16)).float()
        n_run = tower.n_run
        if fold:       # pre_layrnorm + layer 0's layer_norm1 stay explicit; every other LayerNorm rides in a GEMM
            assert calls["ln"] == 2 and calls["fold"] == 4 * n_run - 1, calls
        else:
            assert calls["ln"] == 1 + 2 * n_run and calls["fold"] == 0, calls
        monkeypatch.setattr(ops, "layernorm_fwd_f16", ln0)
        monkeypatch.setattr(ops, "gemm_f16_nt_ln", g0)
    e_fold, e_expl, e_ab = rel(outs[True], ref), rel(outs[False], ref), rel(outs[True], outs[False])
    print(f"select_layer {select_layer}: folded vs oracle {e_fold:.2e}, explicit vs oracle {e_expl:.2e}, folded vs explicit {e_ab:.2e}")
    assert e_fold <= 1.5 * e_expl + 2e-4 and e_fold <= 1e-2 and e_ab <= 1e-2


def test_fold_refuses_columns_that_sink_into_fp16_subnormals_and_falls_back(monkeypatch):
    """ADVICE r3 (clip.py::_fold_prepare): a trained CLIP-L pairs outlier residual channels with very small / large ln_w.  With an
    ln_w spread of 1e-4 .. 10 and a x300 residual channel sitting on a small gain, the folded column W[:, k] ln_w[k] of layer 1's
    q|k|v falls below fp16's normal range: that GEMM must keep its explicit LayerNorm pass (fold_fallbacks names it), every other
    GEMM stays folded, and the tower still matches the fp32 oracle like the explicit tower does.  A spread of 0.05 .. 10 is folded
    everywhere."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import clip as clip_mod
    from hvlm.config import CLIPVisionConfig
    c = dict(clip_d=256, clip_mlp=512, clip_layers=4, clip_heads=4, patch=14, image=28)
    cfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=4, image_size=28, patch_size=14)
    img = synth.make_images(1, 4, 4, 28, seed=19)[0]
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    for lo, expect_fallback in ((1e-4, True), (0.05, False)):
        p = synth.to_torch(synth.clip_params(c, seed=6))
        g = torch.Generator().manual_seed(1)
        for k in list(p):
            if "layer_norm" in k and k.endswith("weight"):
                spread = torch.exp(torch.empty(256).uniform_(float(torch.log(torch.tensor(lo * 20))), float(torch.log(torch.tensor(10.0))), generator=g))
                p[k] = spread
        # an outlier channel (x300 in the residual stream, written by layer 0's fc2) read through a tiny gain by layer 1's LayerNorms
        p["vision_model.encoder.layers.0.mlp.fc2.bias"][17] = 300.0
        p["vision_model.encoder.layers.1.layer_norm1.weight"][17] = lo
        with torch.no_grad():
            ref = R.clip_features(img, p, 4, 4)
        outs = {}
        for fold in (True, False):
            monkeypatch.setattr(clip_mod, "_FOLD_LN", fold)
            tower = clip_mod.CLIPVisionTower(cfg, "cpu", dtype=torch.float16)
            tower.load_state_dict(p)
            outs[fold] = tower(img.to(torch.bfloat16)).float()
            if fold:
                fb = [(i, n) for i, n, _ in tower.fold_fallbacks]
                assert ((1, "qkv") in fb) == expect_fallback, (lo, tower.fold_fallbacks)
                assert len(fb) <= 2, fb                                     # the rest of the tower stays folded
                assert all(tower._pf[f"{i}.{n}_ok"] for i in range(tower.n_run) for n in ("qkv", "fc1") if (i, n) not in fb)
        e_fold, e_expl = rel(outs[True], ref), rel(outs[False], ref)
        print(f"ln_w spread from {lo}: folded (+ fallbacks) vs oracle {e_fold:.2e}, explicit vs oracle {e_expl:.2e}")
        assert torch.isfinite(outs[True]).all()
        assert e_fold <= 1.5 * e_expl + 5e-4 and e_fold <= 2e-2


@pytest.mark.parametrize("fold", [True, False])
def test_last_layer_fc2_after_pooling_equals_pooling_the_full_output(monkeypatch, fold):
    """config.pool_before_last_fc2 (hvlm/clip.py::_last_layer_pooled): pool(x + fc2(h) + b) = pool(x) + fc2(pool(h)) + b — the
    tower called with pool=(B, T) must return what VisualToTokenHelper.compress_tokens (hoi_forecast/model/visual_to_tokens.py:
    252-272; CLS dropped, clip_encoder.py:29-37) makes of its full output, up to the rounding order, and match the fp32 oracle as
    well as the unpooled path does.  Folded and explicit-LayerNorm towers."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from hvlm import clip as clip_mod
    from hvlm import ops
    from hvlm.config import CLIPVisionConfig
    c = dict(clip_d=256, clip_mlp=512, clip_layers=4, clip_heads=4, patch=14, image=56)      # 4 x 4 patches + CLS
    cfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=4, num_attention_heads=4, image_size=56, patch_size=14)
    p = synth.to_torch(synth.clip_params(c, seed=8))
    B, T = 2, 6
    img = synth.make_images(B, T, 3, 56, seed=21)                            # [B, T, 3, 56, 56], duplicate frames inside a sample
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    with torch.no_grad():
        feats = R.clip_features(img.flatten(0, 1), p, 4, 4)                  # [B*T, 16, 256] fp32, CLS dropped
        want = R.temporal_spatial_pool(feats.view(B, T, 16, 256))            # [B, T + 16, 256]
    monkeypatch.setattr(clip_mod, "_FOLD_LN", fold)
    tower = clip_mod.CLIPVisionTower(cfg, "cpu", dtype=torch.float16)
    tower.load_state_dict(p)
    frames = img.flatten(0, 1).to(torch.bfloat16)
    full = tower.forward_tokens(frames)                                       # [B*T, 17, 256] bf16
    ref_path = ops.temporal_spatial_pool_fwd(full.contiguous(), B, T, drop_cls=True, grid=4).float()
    calls = []
    g0 = ops.gemm_f16_nt
    monkeypatch.setattr(ops, "gemm_f16_nt", lambda a, b, **k: (calls.append((a.shape[0], b.shape[0], a.shape[1])), g0(a, b, **k))[1])
    got = tower.forward_tokens(frames, pool=(B, T)).float()
    monkeypatch.setattr(ops, "gemm_f16_nt", g0)
    assert got.shape == (B, T + 16, 256)
    assert (B * (T + 16), 256, 512) in calls                                  # fc2 of the last layer ran on the pooled rows ...
    if not fold:                                                              # ... and on every token only in the layers before it
        assert calls.count((B * T * 17, 256, 512)) == tower.n_run - 1         # (the folded tower's other fc2s go through gemm_f16_nt_ln)
    else:
        assert (B * T * 17, 256, 512) not in calls
    e_new, e_old, e_ab = rel(got, want), rel(ref_path, want), rel(got, ref_path)
    print(f"fold={fold}: pooled-fc2 vs oracle {e_new:.2e}, pool of the full output vs oracle {e_old:.2e}, the two paths {e_ab:.2e}")
    assert e_new <= 1.25 * e_old + 2e-4 and e_ab <= 6e-3
