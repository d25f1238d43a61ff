"""Worker of tests/test_gpu_dp.py: one rank of a 2-rank data-parallel step with the REAL kernels.  Both ranks run on
cuda:0 (the test box has one GPU), so the exchange goes over gloo (HVLM_DIST_BACKEND=gloo; RCCL refuses two ranks on one
device) — everything else is the production path: hvlm.dp.attach hooks fired by LlamaStack.backward, per-bucket AdamW on
the optimizer's own stream ordered after each bucket's all-reduce.
usage: dp_gpu_worker.py RANK WORLD PORT OUT.npz [single]   ("single": no process group, plain backward + dump)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "handsonvlm-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def build(rank):
    import torch
    from oracle import synth
    from test_gpu_model import tiny_model
    m, cfg = tiny_model(training=True)
    c = synth.TINY
    ids, labels, mask = synth.make_text_batch(2, 20 + 4 * rank, seed=3 + rank, n_hand=(4, 4))
    images = synth.make_images(2, c["frames"], 2, c["image"], seed=11 + rank)
    fh, fv = synth.make_hands(2, seed=5 + rank)
    eps = torch.randn(16, 256, generator=torch.Generator().manual_seed(rank)).cuda()
    batch = dict(input_ids=ids, attention_mask=mask, labels=labels, image=images.to(torch.bfloat16), future_hands=fh,
                 contact_point=torch.zeros(2, 2), future_valid=fv, gt_label_valid=torch.ones(2), cvae_eps=eps)
    return m, batch


def dump(m, path, extra):
    import numpy as np
    import torch
    torch.cuda.synchronize()
    out = dict(extra)
    for n, w, g, _ in m.trainable_tensors():
        out["g." + n] = g.float().cpu().numpy()
        out["w." + n] = w.float().cpu().numpy()
    for n, p in m.hand_traj_decoder.named_parameters():
        out["g.head." + n] = p.grad.float().cpu().numpy()
        out["w.head." + n] = p.detach().float().cpu().numpy()
    np.savez(path, **out)


def main():
    rank, world, port, outp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    single = len(sys.argv) > 5 and sys.argv[5] == "single"
    import numpy as np
    import torch
    from hvlm import dp
    from hvlm.optim import AdamW
    if not single:
        os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                          HVLM_DIST_BACKEND="gloo")
        r, l, w = dp.init_from_env()
        assert (r, w) == (rank, world) and torch.distributed.get_backend() == "gloo"
    m, batch = build(rank)
    out = m(**batch)
    if single:
        m.backward()
        dump(m, outp, dict(loss=np.float32(float(out.loss.detach()))))
        return
    opt = AdamW(m, lr=1e-2, overlap=True)
    buckets = dp.attach(m, dp.GradBuckets(), opt)
    assert buckets.active and buckets.world == 2
    opt.begin_step(grad_scale=1.0 / world)
    m.backward()
    buckets.wait()
    torch.cuda.synchronize()
    grads = {n: g.clone() for n, _, g, _ in m.trainable_tensors()}          # the reduced (summed) buckets
    hgr = {n: p.grad.clone() for n, p in m.hand_traj_decoder.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    out_d = dict(loss=np.float32(float(out.loss.detach())), updated_early=np.int32(len(opt._done)))
    for n, w, _, _ in m.trainable_tensors():
        out_d["g." + n] = grads[n].float().cpu().numpy()
        out_d["w." + n] = w.float().cpu().numpy()
    for n, p in m.hand_traj_decoder.named_parameters():
        out_d["g.head." + n] = hgr[n].float().cpu().numpy()
        out_d["w.head." + n] = p.detach().float().cpu().numpy()
    np.savez(outp, **out_d)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
