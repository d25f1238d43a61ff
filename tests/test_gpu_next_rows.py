"""GPU tests of the "next" rows: checkpoint round trip, the DP training loop (world 1), the evaluation
harness, exact frame de-duplication on the EK100 tiling pattern."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import synth  # noqa: E402

BF = torch.bfloat16


def tiny(training=False):
    from test_gpu_model import tiny_model
    return tiny_model(training=training)


def test_checkpoint_roundtrip_bit_exact(tmp_path):
    from hvlm import checkpoint
    m, cfg = tiny()
    m._extra_state = {"time_embedding": torch.randn(20, cfg.hidden_size)}       # unused reference tensor carried through
    checkpoint.save_pretrained(m, str(tmp_path), max_shard_bytes=6 * 2 ** 20)
    m2 = checkpoint.from_pretrained(str(tmp_path), device="cuda:0")
    # the tower is not part of state_dict() (frozen, loaded from its own checkpoint in the reference): copy it over
    m2.tower.p, m2.tower.is_loaded = m.tower.p, True
    ids, _, _ = synth.make_text_batch(1, 24, seed=3)
    img = synth.make_images(1, 100, 2, 224, seed=1).to(BF)
    a = m(input_ids=ids, image=img).logits
    b = m2(input_ids=ids, image=img).logits
    assert torch.equal(a, b)
    assert "time_embedding" in m2._extra_state


def test_dedup_exact_on_epic_tiling():
    from hvlm.data import tile_epic_frames
    m, cfg = tiny()
    f = torch.randn(2, 10, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(BF)
    img = torch.stack([tile_epic_frames(f[0]), tile_epic_frames(f[1])]).cuda()
    a = m.visual_tokens(img)
    m.config.dedup_frames = True
    b = m.visual_tokens(img)
    assert torch.equal(a, b)
    img2 = img.clone(); img2[1, 57, 0, 0, 0] += 1.0                            # one perturbed frame: still exact
    m.config.dedup_frames = False
    c = m.visual_tokens(img2)
    m.config.dedup_frames = True
    assert torch.equal(c, m.visual_tokens(img2))


def test_train_loop_and_resume(tmp_path):
    """hvlm.train.train on the real kernels: (1) a fixed batch is learnt; (2) auto-resume (handsonvlm/train/train.py:399-402)
    restores weights, fp32 masters / moments / step count, the CVAE head's optimizer, the schedule position, the data
    order and the RNG: an interrupted-and-resumed run ends with BIT-identical weights and optimizer state (the step has no
    atomics: embedding gradient by sorted segments, column sums and the CE loss in fixed order)."""
    import os
    from hvlm.data import SyntheticEpicDataset
    from hvlm.train import train
    torch.manual_seed(0); torch.cuda.manual_seed_all(0)                          # the CVAE noise of every step
    m, cfg = tiny(training=True)
    m.config.dedup_frames = True
    ds = SyntheticEpicDataset(2, min_len=50, max_len=60)                        # the same two samples every step
    log = train(m, ds, output_dir=str(tmp_path / "fixed"), per_device_batch=2, epochs=1, max_steps=8, lr=3e-4, log_every=1,
                warmup_ratio=0.0, save_every_epoch=False)
    assert len(log) == 8 and all(np.isfinite(r["loss"]) for r in log)
    tl = [r["text loss"] for r in log]
    # fixed batch: CE falls; the trunk also receives the CVAE head's gradient (fresh noise every step), so one of the
    # seven transitions may go the other way (seen once in ten unseeded runs)
    assert sum(b < a for a, b in zip(tl, tl[1:])) >= 6 and tl[-1] < tl[0] - 0.1
    assert os.path.isdir(tmp_path / "fixed" / "checkpoint-8" / "optimizer")

    ds6 = SyntheticEpicDataset(6, min_len=50, max_len=60)                       # 3 batches per epoch: mid-epoch interruption
    kw = dict(per_device_batch=2, epochs=1, max_steps=8, lr=3e-3, log_every=1, warmup_ratio=0.25, save_every_epoch=False,
              save_steps=4, seed=5)

    def snapshot(model, d):
        w = {n: t.clone() for n, t, _, _ in model.trainable_tensors()}
        w.update({"wt." + n: t.clone() for n, _, _, t in model.trainable_tensors() if t is not None})
        o = {f: torch.load(os.path.join(d, f), map_location="cpu", weights_only=True) for f in sorted(os.listdir(d))}
        return w, o

    torch.manual_seed(1); torch.cuda.manual_seed_all(1)
    ma, _ = tiny(training=True); ma.config.dedup_frames = True
    full = train(ma, ds6, output_dir=str(tmp_path / "a"), **kw)
    wa, oa = snapshot(ma, str(tmp_path / "a" / "checkpoint-8" / "optimizer"))
    torch.manual_seed(1); torch.cuda.manual_seed_all(1)
    mb, _ = tiny(training=True); mb.config.dedup_frames = True
    part = train(mb, ds6, output_dir=str(tmp_path / "b"), stop_after=6, **kw)    # "killed" after step 6, last save at 4
    assert [r["step"] for r in part] == [1, 2, 3, 4, 5, 6]
    torch.manual_seed(999); torch.cuda.manual_seed_all(999)                      # a new process has unrelated RNG state
    mc, _ = tiny(training=True); mc.config.dedup_frames = True                   # ... and pre-training weights
    rest = train(mc, ds6, output_dir=str(tmp_path / "b"), **kw)
    assert [r["step"] for r in rest] == [5, 6, 7, 8]
    assert [r["loss"] for r in rest] == [r["loss"] for r in full[4:]]            # continues, bit for bit
    assert rest[0]["loss"] != full[0]["loss"]                                    # ... and did not jump back to step 1
    wc, oc = snapshot(mc, str(tmp_path / "b" / "checkpoint-8" / "optimizer"))
    for k in wa:
        assert torch.equal(wa[k], wc[k]), k
    for f in oa:
        if f == "meta.pt":
            assert oa[f]["step_count"] == oc[f]["step_count"] == 8
            continue
        for n in oa[f]:
            for kk in ("master", "m", "v"):
                assert torch.equal(oa[f][n][kk], oc[f][n][kk]), (f, n, kk)


def test_evaluation_harness_metrics_and_failed_generations():
    from hvlm import evaluate as E
    from hvlm.model import SampleDecoderOnlyOutput
    m, cfg = tiny()
    r = np.random.RandomState(0)
    samples = [dict(input_ids=torch.zeros(1, 4, dtype=torch.long), image=None,
                    future_hands=torch.from_numpy(r.rand(1, 2, 5, 2).astype(np.float32)), future_valid=torch.ones(1, 2))
               for _ in range(5)]
    preds = [torch.from_numpy(r.rand(1, 2, 4, 2).astype(np.float32)) for _ in range(5)]
    it = iter(range(5))

    def fake_generate(input_ids, image=None, **kw):
        i = next(it)
        ph = preds[i] if i != 3 else torch.empty(1, 2, 0, 2)                    # sample 3: generation without hand tokens
        return SampleDecoderOnlyOutput(sequences=input_ids, pred_hands=ph)

    m.generate = fake_generate
    res = E.evaluate_traj(m, samples)
    from oracle import ref_cpu as R
    P = np.stack([p.numpy() if i != 3 else np.zeros((1, 2, 4, 2), np.float32) for i, p in enumerate(preds)])
    G = np.stack([s["future_hands"][0, :, 1:].numpy() for s in samples])
    V = np.ones((5, 2), np.float32); V[3] = 0
    a, f, w = R.evaluate_traj_stochastic(P, G, V)
    assert np.allclose([res["ade"], res["fde"], res["wde"]], [a, f, w], rtol=1e-6) and res["n"] == 5


def test_fp8_forward_close_to_bf16_forward():
    """BASELINE config 5 (fp8 MFMA forward): the decoder stack with e4m3 QKV/O/FFN GEMMs (per-tensor static scales,
    calibrated on the input) against the bf16 forward of the same weights.  e4m3 keeps 3 mantissa bits, so this is a
    closeness check (stated bound), not bit parity; the fp8 GEMM itself is pinned exactly in test_gpu_ops.py."""
    import torch
    from hvlm.config import HandsOnVLMConfig
    from hvlm.llama import LlamaStack
    cfg = HandsOnVLMConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4)
    st = LlamaStack(cfg, device="cuda:0", training=False)
    g = torch.Generator(device="cuda").manual_seed(0)
    for lp in st.layers:
        for k in ("qkv", "o", "gu", "down"):
            lp.w[k].copy_((torch.randn(lp.w[k].shape, device="cuda", generator=g) * 0.03).to(torch.bfloat16))
        lp.w["ln1"].fill_(1.0); lp.w["ln2"].fill_(1.0)
    st.norm_w.fill_(1.0)
    B, S = 2, 256
    x = (torch.randn(B * S, 512, device="cuda", generator=g)).to(torch.bfloat16)
    ref = st.forward(x, B, S, save=False).float()
    st.fp8_prepare(x, B, S)
    got = st.forward_fp8(x, B, S).float()
    r = float((got - ref).norm() / ref.norm())
    print("fp8 vs bf16 forward rel", r)
    assert float(ref.norm()) > 1.0 and 1e-4 < r < 8e-2 and torch.isfinite(got).all()


def test_overlapped_optimizer_bit_identical_to_end_of_step_update():
    """AdamW per gradient bucket on the optimizer stream, started from the backward hooks (optim.AdamW overlap=True),
    against the same update done after the whole backward: every weight, transposed copy and moment bit-identical after
    two steps (the reference does not clip gradients, so no update depends on the other buckets)."""
    from test_gpu_model import tiny_model, tiny_batch
    from hvlm import dp
    from hvlm.optim import AdamW
    ids, labels, mask, images, fh, fv = tiny_batch()
    eps = torch.randn(2 * 8, 256, generator=torch.Generator().manual_seed(1)).cuda()
    snaps = []
    for overlap in (False, True):
        m, cfg = tiny_model(training=True)
        opt = AdamW(m, lr=1e-3, weight_decay=0.01, overlap=overlap)
        buckets = dp.attach(m, dp.GradBuckets(), opt)
        assert opt.overlap == overlap and set(opt.groups) == {f"layers.{i}" for i in range(cfg.num_hidden_layers)} | {"head", "tail"}
        for _ in range(2):
            out = m(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(BF), future_hands=fh.clone(),
                    future_valid=fv.clone(), cvae_eps=eps)
            opt.begin_step(grad_scale=0.5)
            out.loss.backward()
            if overlap:
                assert opt._done == set(opt.groups)                 # every bucket was handed over from a hook
            buckets.wait()
            opt.step()
        torch.cuda.synchronize()
        snaps.append([(st["name"], st["w"].clone(), None if st["wt"] is None else st["wt"].clone(), st["m"].clone(),
                       st["v"].clone(), st["master"].clone()) for st in opt.state])
    for a, b in zip(*snaps):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert (x is None and y is None) or torch.equal(x, y), a[0]


def test_reference_written_checkpoint_on_hip(tmp_path):
    """SURVEY.md §8f-1 on the real kernels: tests/golden/ckpt_tiny (written from the reference model: its key names, its
    config.json, transformers-4.31 .bin shards) -> checkpoint.from_pretrained -> the reference's logits / loss, and the
    LLaVA `mm_projector.bin` adapter override (llava_arch.py:64-69)."""
    import os
    from test_ckpt_reference_cpu import batch, check_loaded_model, inflate, rel
    from hvlm import checkpoint
    d = inflate(str(tmp_path / "ck"))
    exp = dict(np.load(os.path.join(d, "expected.npz")))
    m = checkpoint.from_pretrained(d, device="cuda:0", training=False)
    assert m.tower.is_loaded and m.config.vision.hidden_size == 64
    check_loaded_model(m, exp, "cuda")
    m2 = checkpoint.from_pretrained(os.path.join(d, "st"), device="cuda:0", training=False,
                                    mm_projector_bin=os.path.join(d, "mm_projector.bin"))
    ids, labels, mask, images, fh, fv = batch()
    out = m2(input_ids=ids, attention_mask=mask, image=images.to(BF), is_evaluate=True)
    valid = torch.ones(2, 379, dtype=torch.bool); valid[1, 374:] = False
    got = out.logits.float().cpu()[:, ::7, ::251][valid[:, ::7]]
    assert rel(got, exp["logits_slice_adapter"][valid[:, ::7].numpy()]) < 1.5e-2


def test_entry_points_on_hip(tmp_path, capsys):
    """SURVEY.md §8 f1 + f2 joined, on the real kernels: `python -m hvlm.train` with the command line of scripts/finetune.sh loads
    tests/golden/ckpt_tiny through hvlm.checkpoint (handsonvlm/train/train.py:235-240), trains two steps, writes a self-contained
    checkpoint (tower included, like the reference's full state dict), and `python -m hvlm.evaluate --model-path` loads it back
    (handsonvlm/evaluation/evaluate.py:18-26).  The CPU twin with the ops stand-in is tests/test_entrypoints_cpu.py."""
    import json
    import os
    from test_ckpt_reference_cpu import inflate
    from test_entrypoints_cpu import FINETUNE_SH_FLAGS
    from hvlm import checkpoint, evaluate, train
    d = inflate(str(tmp_path / "ck"))
    out = str(tmp_path / "out")
    torch.manual_seed(0)
    log = train.main(["--model_name_or_path", d, "--pretrain_mm_mlp_adapter", os.path.join(d, "mm_projector.bin"), "--output_dir", out,
                      "--num_train_epochs", "1", "--per_device_train_batch_size", "2", "--samples_per_epoch", "4"] + FINETUNE_SH_FLAGS)
    assert [r["step"] for r in log][-1] == 2 and all(np.isfinite(r["loss"]) for r in log)
    ck = os.path.join(out, "checkpoint-2")
    sd = checkpoint.load_state(ck)
    assert any(k.startswith("model.vision_tower.vision_tower.") for k in sd)           # self-contained
    assert sd["model.embed_tokens.weight"].shape[0] == 32101
    capsys.readouterr()
    res = evaluate.main(["--model-path", ck, "--synthetic", "3", "--batch-size", "3", "--max-new-tokens", "4"])
    assert set(res) == {"ade", "fde", "wde", "n"} and res["n"] == 3
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["n"] == 3
    # the same evaluation in e4m3 (BASELINE config 5's switches through the entry point)
    res8 = evaluate.main(["--model-path", ck, "--synthetic", "3", "--batch-size", "3", "--max-new-tokens", "4", "--fp8"])
    assert res8["n"] == 3
