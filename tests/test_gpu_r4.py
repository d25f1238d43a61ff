"""Round-4 GPU parity tests of the exact-algebra switch `config.pool_before_last_fc2` (round-3 verdict item 5a):

the LAST CLIP layer's fc2 + residual run AFTER the temporal/spatial pooling.  Reference order: CLIPEncoderLayer (transformers
modeling_clip, through llava/model/multimodal_encoder/clip_encoder.py:39-51) on every token, then
VisualToTokenHelper.compress_tokens (hoi_forecast/model/visual_to_tokens.py:252-272).  The checker is the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oracle import synth  # noqa: E402

BF, F16 = torch.bfloat16, torch.float16


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", [F16, BF])
@pytest.mark.parametrize("B,T,G,D", [(2, 100, 16, 1024), (1, 100, 16, 4096), (3, 7, 24, 256)])
def test_pool_kernel_fp16_and_bf16_vs_oracle(dtype, B, T, G, D):
    """hvlm_temporal_spatial_pool_fwd[_f16] on [B*T, G*G+1, D] tokens (CLS dropped): fp32 sums of the 16-bit values, one
    rounding — equal to the oracle's fp32 pooling rounded to the same dtype up to the summation order (<= 1 ulp)."""
    from hvlm import ops
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + D)
    tok = (torch.randn(B * T, G * G + 1, D, device="cuda", generator=g) * 2.0 + 0.3).to(dtype)
    got = ops.temporal_spatial_pool_fwd(tok, B, T, drop_cls=True, grid=G)
    assert got.dtype == dtype and got.shape == (B, T + G * G, D)
    want = R.temporal_spatial_pool(tok.float().cpu().view(B, T, G * G + 1, D)[:, :, 1:])
    ulp = 2.0 ** (-10 if dtype == F16 else -7)
    err = (got.float().cpu() - want).abs()
    assert float((err / want.abs().clamp(min=1e-3)).max()) <= 1.01 * ulp
    assert torch.equal(got, ops.temporal_spatial_pool_fwd(tok, B, T, drop_cls=True, grid=G))


def test_vit_l14_tower_last_fc2_after_pooling_vs_oracle():
    """ViT-L/14 fp16 tower, one sample of 100 frames (10 distinct, tiled like EK100): forward_tokens(pool=(1, 100)) against
    (i) the fp32 oracle's features pooled by the oracle, (ii) the library's own default path (every token through the last fc2,
    then the pool kernel).  The pooled path must be at least as close to the oracle as the default (+ 25 %), the two paths
    within bf16 rounding of each other."""
    from hvlm import ops
    from hvlm.clip import CLIPVisionTower
    from hvlm.config import CLIPVisionConfig
    c = dict(clip_d=1024, clip_mlp=4096, clip_layers=24, clip_heads=16, patch=14, image=224)
    pnp = synth.clip_params(c, seed=77)
    p = synth.to_torch(pnp)
    r = np.random.RandomState(5)
    base = torch.from_numpy(r.standard_normal((10, 3, 224, 224)).astype(np.float32)).to(BF)
    frames = base[torch.arange(100) % 10].contiguous()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    with torch.no_grad():
        f10 = R.clip_features(base.float(), p, 24, 16)                         # fp32, [10, 256, 1024]
        want = R.temporal_spatial_pool(f10[torch.arange(100) % 10][None])      # [1, 356, 1024]
    tower = CLIPVisionTower(CLIPVisionConfig(), "cuda:0", dtype=F16)
    tower.load_state_dict(p)
    full = tower.forward_tokens(frames.cuda())                                 # [100, 257, 1024] bf16
    old = ops.temporal_spatial_pool_fwd(full.contiguous(), 1, 100, drop_cls=True, grid=16)
    new = tower.forward_tokens(frames.cuda(), pool=(1, 100))
    assert new.shape == (1, 356, 1024) and new.dtype == BF
    e_new, e_old, e_ab = rel(new, want), rel(old, want), rel(new, old)
    print(f"ViT-L/14: pooled-fc2 vs fp32 oracle {e_new:.3e}, default path {e_old:.3e}, the two paths {e_ab:.3e}")
    assert e_new <= 1.25 * e_old and e_new <= 6e-3 and e_ab <= 6e-3
    assert torch.equal(new, tower.forward_tokens(frames.cuda(), pool=(1, 100)))


def test_model_switch_changes_nothing_but_rounding(golden):
    """HandsOnVLMForCausalLM with pool_before_last_fc2 on / off on the reference-golden batch: same loss and gradients up to bf16
    rounding, both inside the reference golden's bounds (handsonvlm.py:70-210 through the fixtures of tests/test_gpu_model.py)."""
    from test_gpu_model import tiny_batch, tiny_model
    g3 = golden("g3_forward.npz")
    ids, labels, mask, images, fh, fv = tiny_batch()
    res = {}
    for flag in (False, True):
        m, cfg = tiny_model(pool_before_last_fc2=flag)
        calls = []
        from hvlm import ops
        g0 = ops.gemm_f16_nt
        ops.gemm_f16_nt = lambda a, b, **k: (calls.append(a.shape[0]), g0(a, b, **k))[1]
        try:
            out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(BF), future_hands=fh.clone(),
                    contact_point=torch.zeros(2, 2), future_valid=fv.clone(), gt_label_valid=torch.ones(2),
                    cvae_eps=torch.from_numpy(g3["eps_noise"]).cuda())
        finally:
            ops.gemm_f16_nt = g0
        assert (2 * 356 in calls) == flag                                      # fc2 on the pooled rows of both samples
        loss = float(out.loss.detach())
        out.loss.backward()
        res[flag] = (loss, {n: gr.float().clone() for n, _, gr, _ in m.trainable_tensors()})
    (l0, g0_), (l1, g1_) = res[False], res[True]
    assert abs(l0 - l1) / abs(l0) < 2e-3
    assert abs(l1 - float(g3["loss"])) / float(g3["loss"]) < 2e-3
    for n in ("proj_w", "proj_b", "layers.0.qkv", "embed"):
        assert rel(g1_[n], g0_[n]) < 2e-2, (n, rel(g1_[n], g0_[n]))
    assert rel(g1_["proj_w"], g3["g_proj_w"]) < 3e-2 and rel(g1_["proj_b"], g3["g_proj_b"]) < 3e-2


@pytest.mark.parametrize("N,HW,P", [(3, 224, 14), (2, 336, 14), (2, 28, 14), (2, 35, 7)])
def test_im2col_strip_kernel_exact(N, HW, P):
    """hvlm_patch_im2col (coalesced strip kernel for even P / HW, the element-wise kernel otherwise) and hvlm_patch_im2col_f16 (the
    frames' cast to the tower's fp16 applied on the way through, clip_encoder.py:48): a pure gather — bit-exact against the unfold
    of the frames, the fp16 form equal to casting the frames first; the K padding is zero."""
    from hvlm import ops
    g = torch.Generator(device="cuda").manual_seed(N * HW)
    img = (torch.randn(N, 3, HW, HW, device="cuda", generator=g) * 2.0).to(BF)
    G, K = HW // P, 3 * P * P
    kpad = (K + 63) // 64 * 64
    want = img.reshape(N, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(N * G * G, K)
    got = ops.patch_im2col(img, P, kpad)
    assert got.shape == (N * G * G, kpad) and torch.equal(got[:, :K], want) and float(got[:, K:].abs().max() if kpad > K else 0.0) == 0.0
    if P % 2 == 0 and HW % 2 == 0:
        got16 = ops.patch_im2col(img, P, kpad, to_f16=True)
        assert got16.dtype == F16 and torch.equal(got16[:, :K], want.to(F16)) and float(got16[:, K:].abs().max() if kpad > K else 0.0) == 0.0
        assert torch.equal(got16, ops.patch_im2col(ops.cast16(img, F16).view(BF), P, kpad).view(F16))
