"""Host logic of the decoder engine on the ops stand-in (tests/cpu_ops.py, test infrastructure):
  * activation recompute (config.activation_recompute; the reference's gradient checkpointing, scripts/finetune.sh:36)
    gives BIT-identical loss and gradients while keeping only each layer's input between forward and backward;
  * packed (unpadded, cu_seqlens) batches (config.pack_sequences; the reference's flash-attn patch contract,
    llama_flash_attn_monkey_patch.py:81-96) give the padded batch's loss and gradients (up to fp32 summation order)."""
import numpy as np
import pytest
import torch

from oracle import synth


def _model(**cfg_kw):
    from test_train_resume_cpu import _tiny
    m, cfg = _tiny()
    m.config.dedup_frames = False
    for k, v in cfg_kw.items():
        setattr(m.config, k, v)
    ar = m.config.activation_recompute
    m.stack.recompute = m.config.num_hidden_layers if ar is True else int(ar or 0)
    return m


def _batch(ragged=True):
    c = synth.TINY
    ids, labels, mask = synth.make_text_batch(3, 28, seed=3, n_hand=(4, 0, 4), pad_tail=(0, 9, 4) if ragged else (0, 0, 0))
    # the collator's padding is at the END of input_ids; cut each sample to its own length by marking pads ignored
    images = synth.make_images(3, c["frames"], 2, c["image"], seed=11).to(torch.bfloat16)
    fh, fv = synth.make_hands(3, seed=5)
    eps = torch.randn(24, 256, generator=torch.Generator().manual_seed(0))
    return dict(input_ids=ids, attention_mask=mask, labels=labels, image=images, future_hands=fh, contact_point=torch.zeros(3, 2),
                future_valid=fv, gt_label_valid=torch.ones(3), cvae_eps=eps)


def _run(m, batch):
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    out = m(**b)
    saved_keys = [sorted(sv.keys()) for sv in m.stack.saved]
    out.loss.backward()
    g = {n: gr.clone() for n, _, gr, _ in m.trainable_tensors()}
    g.update({"head." + n: p.grad.clone() for n, p in m.hand_traj_decoder.named_parameters()})
    return float(out.loss.detach()), g, saved_keys, out


def test_activation_recompute_is_bit_identical(monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    batch = _batch()
    l0, g0, k0, _ = _run(_model(), batch)
    l1, g1, k1, _ = _run(_model(activation_recompute=True), batch)
    assert all(k == ["x"] for k in k1) and all(len(k) > 5 for k in k0)        # only the layer inputs were kept
    assert l0 == l1
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    l2, g2, k2, _ = _run(_model(activation_recompute=1), batch)               # first layer only
    assert k2[0] == ["x"] and len(k2[1]) > 5 and l2 == l0 and all(torch.equal(g0[n], g2[n]) for n in g0)


def test_packed_batch_matches_padded_batch(monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch)
    batch = _batch()
    # trim the collator padding: samples 1 and 2 end 9 / 4 tokens early -> their ids are shorter in the packed plan only
    # if the pads are removed from input_ids; the reference pads with pad_token_id and masks — keep both paths identical by
    # feeding per-sample lengths through attention_mask-free padding ids (pad id 0 rows are real rows for both engines).
    l0, g0, _, out0 = _run(_model(), batch)
    mp = _model(pack_sequences=True)
    l1, g1, _, out1 = _run(mp, batch)
    assert out1.cu_seqlens is not None and out1.logits.dim() == 2
    assert abs(l0 - l1) / abs(l0) < 2e-4, (l0, l1)
    for n in g0:
        a, b = g0[n].float(), g1[n].float()
        r = float((a - b).norm() / (a.norm() + 1e-30))
        assert r < 2e-2, (n, r)
    # packed logits == the valid rows of the padded logits
    cu = out1.cu_seqlens.tolist()
    # (out0.logits was overwritten in place by backward: compare a fresh forward)
    m0 = _model()
    b0 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    lg0 = m0(**b0, keep_logits=True).logits
    b1 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    lg1 = mp(**b1, keep_logits=True).logits
    for b in range(3):
        n = cu[b + 1] - cu[b]
        d = (lg0[b, :n].float() - lg1[cu[b]:cu[b + 1]].float()).abs().max()
        assert float(d) <= 0.05 * float(lg0[b, :n].float().abs().max()), (b, float(d))


def test_packed_plan_layout(monkeypatch):
    """Pure host logic: the packed plan is the padded plan with the collator's padding rows removed."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from test_train_resume_cpu import _tiny
    m, cfg = _tiny()
    ids, labels, mask = synth.make_text_batch(3, 28, seed=3, n_hand=(4, 0, 4), pad_tail=(0, 9, 4))
    pa = m.build_splice_plan(ids, labels, 356, 4, False, True)
    pk = m.build_splice_plan(ids, labels, 356, 4, False, True, pack=True, attention_mask=mask)
    S = pa.S
    lens = [S, S - 9, S - 4]
    assert pk.cu.tolist() == [0, lens[0], lens[0] + lens[1], sum(lens)] and pk.M == sum(lens) and pk.S == S
    for b in range(3):
        n, o = lens[b], int(pk.cu[b])
        assert torch.equal(pk.src[o:o + n], pa.src.view(3, S)[b, :n])
        assert torch.equal(pk.hand_slot[o:o + n], pa.hand_slot.view(3, S)[b, :n])
        assert torch.equal(pk.targets[o:o + n - 1], pa.targets.view(3, S)[b, :n - 1]) and int(pk.targets[o + n - 1]) == -100
        assert torch.equal(pk.pos[o:o + n], torch.arange(n, dtype=torch.int32)) and (pk.row_sample[o:o + n] == b).all()
        hr_pad = pa.hand_rows.view(3, 4)[b]
        want = torch.where(hr_pad >= 0, hr_pad - b * S + o, hr_pad)
        assert torch.equal(pk.hand_rows.view(3, 4)[b], want)
    assert torch.equal(pk.labels, pa.labels)                   # the padded labels the reference returns


def test_sample_batch_equals_per_sample_sampling(monkeypatch):
    """model.sample_batch (batched prefill into a [B, max_len, 3D] cache + one batched decode step per token) reproduces,
    sample by sample, what the reference-shaped batch-1 sample() loop produces (forced tokens, pinned CVAE noise): same
    sequences, same hands — with different prompt lengths (right padding) and different hand counts per sample."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from test_train_resume_cpu import _tiny
    m, cfg = _tiny(training=False)
    m.config.dedup_frames = False
    c = synth.TINY
    ids, _, mask = synth.make_text_batch(3, 16, seed=8, n_hand=(0, 0, 0), pad_tail=(0, 5, 2))
    images = synth.make_images(3, c["frames"], 2, c["image"], seed=12).to(torch.bfloat16)
    forced = [[512, 32100, 32100, 77, 2], [32100, 9, 32100, 32100, 2], [600, 601, 2, 2, 2]]
    T = ids.shape[1]
    lens0 = [16, 11, 14]
    g = torch.Generator().manual_seed(4)
    noise = [[2.0 * torch.randn(2, 256, generator=g) for _ in range(4)] for _ in range(3)]

    def force_batch(cur, scores):
        out = torch.full_like(scores, -1e9)
        step = cur.shape[1] - T
        for b in range(3):
            out[b, forced[b][min(step, 4)]] = 0.0
        return out

    ob = m.sample_batch(ids, logits_processor=[force_batch], max_length=T + 5, eos_token_id=2, attention_mask=mask, image=images,
                        hand_noise=noise)
    for b in range(3):
        prompt = ids[b:b + 1, : lens0[b]]

        def force_one(cur, scores, b=b, n0=lens0[b]):
            out = torch.full_like(scores, -1e9)
            out[:, forced[b][cur.shape[1] - n0]] = 0.0
            return out

        o1 = m.sample(prompt, logits_processor=[force_one], max_length=lens0[b] + 5, eos_token_id=2, image=images[b:b + 1],
                      hand_noise=noise[b], use_kv_cache=False)
        n = ob.lengths[b]
        assert ob.sequences[b, :n].tolist() == o1.sequences[0].tolist(), b
        assert ob.num_hands[b] == o1.pred_hands.shape[2]
        if ob.num_hands[b]:
            a, r = ob.pred_hands[b, :, : ob.num_hands[b]].float(), o1.pred_hands[0].float()
            assert float((a - r).abs().max()) <= 2e-2 * float(r.abs().max()), (b, a, r)
    assert ob.num_hands == [2, 3, 0] and ob.lengths == [21, 16, 17]


def test_batched_generation_with_model_chosen_tokens(monkeypatch):
    """Greedy decoding where the MODEL picks the tokens (only <hand_traj> is boosted at two steps so that hands get decoded):
    sample_batch on right-padded prompts == the reference-shaped one-by-one loop, token for token and hand for hand."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    from test_train_resume_cpu import _tiny
    m, cfg = _tiny(training=False)
    m.config.dedup_frames = False
    c = synth.TINY
    lens = (12, 9, 14)
    prompts, imgs = [], []
    for i, T in enumerate(lens):
        ids, _, _ = synth.make_text_batch(1, T, seed=20 + i, n_hand=(0,))
        prompts.append(ids[0])
        imgs.append(synth.make_images(1, c["frames"], 2, c["image"], seed=30 + i).to(torch.bfloat16))
    g = torch.Generator().manual_seed(1)
    noise = [[2.0 * torch.randn(2, 256, generator=g) for _ in range(8)] for _ in range(3)]
    ids_b = torch.nn.utils.rnn.pad_sequence(prompts, batch_first=True)
    mask = torch.nn.utils.rnn.pad_sequence([torch.ones(n, dtype=torch.bool) for n in lens], batch_first=True)
    T = ids_b.shape[1]
    greedy = lambda ids, s: torch.full_like(s, float("-inf")).scatter(1, s.argmax(-1, keepdim=True), 0.0)  # noqa: E731

    def boost_batch(cur, scores):
        if cur.shape[1] - T in (1, 3):
            scores = scores.clone(); scores[:, 32100] += 1e4
        return scores

    ob = m.sample_batch(ids_b, logits_processor=[boost_batch, greedy], max_length=T + 6, eos_token_id=2, attention_mask=mask,
                        image=torch.cat(imgs), hand_noise=noise)
    assert ob.num_hands == [2, 2, 2]
    for b in range(3):
        def boost_one(cur, scores, n0=lens[b]):
            if cur.shape[1] - n0 in (1, 3):
                scores = scores.clone(); scores[:, 32100] += 1e4
            return scores
        o1 = m.sample(prompts[b][None], logits_processor=[boost_one, greedy], max_length=lens[b] + 6, eos_token_id=2, image=imgs[b],
                      hand_noise=noise[b], use_kv_cache=False)
        assert ob.sequences[b, : ob.lengths[b]].tolist() == o1.sequences[0].tolist(), b
        assert torch.allclose(ob.pred_hands[b, :, :2].float(), o1.pred_hands[0].float(), atol=2e-2)


def test_evaluate_traj_batched_plumbing(monkeypatch):
    """evaluate_traj(batch_size=2): samples are right-padded into batches of two (last batch partial), generate_batch is
    called per batch and its hands are scored exactly as the one-by-one path scores them."""
    from hvlm import evaluate as E
    from hvlm.model import SampleDecoderOnlyOutput
    r = np.random.RandomState(0)
    samples = [dict(input_ids=torch.arange(3 + i)[None], image=torch.zeros(1, 1), future_hands=torch.from_numpy(r.rand(1, 2, 5, 2).astype(np.float32)),
                    future_valid=torch.ones(1, 2)) for i in range(5)]
    preds = [torch.from_numpy(r.rand(1, 2, 4, 2).astype(np.float32)) for _ in range(5)]
    calls = []

    class Fake:
        def generate(self, input_ids, image=None, **kw):
            i = input_ids.shape[1] - 3
            ph = preds[i] if i != 3 else torch.empty(1, 2, 0, 2)
            return SampleDecoderOnlyOutput(sequences=input_ids, pred_hands=ph)

        def generate_batch(self, input_ids, image=None, attention_mask=None, **kw):
            n = attention_mask.sum(1).tolist()
            calls.append(n)
            B = len(n)
            ph = torch.zeros(B, 2, 4, 2)
            nh = []
            for b in range(B):
                i = n[b] - 3
                assert input_ids[b, : n[b]].tolist() == list(range(n[b])) and (input_ids[b, n[b]:] == 0).all()
                if i != 3:
                    ph[b] = preds[i][0]
                nh.append(0 if i == 3 else 4)
            o = SampleDecoderOnlyOutput(sequences=input_ids, pred_hands=ph)
            o.num_hands = nh
            return o

    a = E.evaluate_traj(Fake(), samples)
    b = E.evaluate_traj(Fake(), samples, batch_size=2)
    assert calls == [[3, 4], [5, 6], [7]] and a == b and a["n"] == 5 and np.isfinite(a["ade"])


def test_lm_head_on_labelled_rows_only_matches_full_logits(monkeypatch):
    """config.lm_head_labelled_rows: lm_head, the cross-entropy and their backward on the rows that carry a target.  The other rows
    (visual tokens, prompt, padding) have zero loss and zero gradient in the reference (CrossEntropyLoss ignore_index = -100 over
    ALL positions, handsonvlm.py:132-145), so loss and every gradient must equal the full-logits pass up to the fp32 order of the
    sums over rows; padded and packed layouts; reading `out.logits` raises (naming the switch) unless keep_logits=True, and
    `out.logits_rows` / `out.rows` are the compact logits."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    batch = _batch()
    for pack in (False, True):
        l0, g0, _, out0 = _run(_model(pack_sequences=pack), batch)
        m1 = _model(pack_sequences=pack, lm_head_labelled_rows=True)
        calls = []
        from hvlm import ops
        g_nt = ops.gemm_nt
        monkeypatch.setattr(ops, "gemm_nt", lambda a, b, *k, **kw: (calls.append((a.shape[0], b.shape[0])), g_nt(a, b, *k, **kw))[1])
        l1, g1, _, out1 = _run(m1, batch)
        monkeypatch.setattr(ops, "gemm_nt", g_nt)
        assert out0.logits is not None
        with pytest.raises(RuntimeError, match="lm_head_labelled_rows"):
            out1.logits
        Vp = m1.lm_head.shape[0]
        rows = [r for r, n in calls if n == Vp]
        n_valid = int((batch["labels"][:, 1:] != -100).sum())
        assert rows == [n_valid], (rows, n_valid)                     # the lm_head GEMM ran on exactly the labelled rows
        assert out1.rows.numel() == n_valid and out1.logits_rows.shape == (n_valid, m1.config.vocab_size)
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (l0, l1)
        for n in g0:
            a, b = g0[n].float(), g1[n].float()
            r = float((a - b).norm() / (a.norm() + 1e-30))
            assert r < 2e-3, (pack, n, r)
        # keep_logits=True hands out the full logits and takes the full path
        b2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        assert m1(**b2, keep_logits=True).logits is not None
        # output_hidden_states=True: every layer's full hidden state is handed out, so the last layer is NOT pruned — only lm_head and
        # the cross-entropy run on the labelled rows; same loss and gradients again
        b3 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        m3 = _model(pack_sequences=pack, lm_head_labelled_rows=True)
        out3 = m3(**b3, output_hidden_states=True)
        assert out3.logits_rows is not None and len(out3.hidden_states) == m3.config.num_hidden_layers + 1
        assert all("rows" not in sv for sv in m3.stack.saved)
        out3.loss.backward()
        assert abs(float(out3.loss.detach()) - l0) <= 2e-6 * abs(l0)
        g3 = {n: gr for n, _, gr, _ in m3.trainable_tensors()}
        for n in g3:
            a, b = g0[n].float(), g3[n].float()
            assert float((a - b).norm() / (a.norm() + 1e-30)) < 2e-3, (pack, n)


def test_labelled_rows_with_activation_recompute_is_bit_identical(monkeypatch):
    """The last layer's row-pruned form (o_proj and MLP on the labelled rows only) is re-run identically by the recompute path."""
    import cpu_ops
    cpu_ops.install(monkeypatch)
    batch = _batch()
    l0, g0, k0, _ = _run(_model(lm_head_labelled_rows=True), batch)
    l1, g1, k1, _ = _run(_model(lm_head_labelled_rows=True, activation_recompute=True), batch)
    assert "rows" in k0[-1] and "ao_v" in k0[-1] and all("rows" not in k for k in k0[:-1])     # only the LAST layer is pruned
    assert all(k == ["x"] for k in k1)
    assert l0 == l1
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
