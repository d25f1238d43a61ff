"""World-size-2 `gloo` test of the data-parallel gradient exchange (hvlm.dp): flat per-layer buckets are
summed across ranks asynchronously from the backward hooks and the optimizer's grad_scale=1/world turns
the sum into the mean — the reference's (DeepSpeed) gradient averaging, without ZeRO."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hvlm import dp
    r, l, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)

    class Layer:
        def __init__(self, i):
            self.flat_g = torch.full((1000,), float(rank + 1) * (i + 1))

    class Stack:
        layers = [Layer(0), Layer(1), Layer(2)]
        norm_g = torch.full((8,), float(rank))
        on_layer_grad_ready = None

    class Head(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(4))

    class Model:
        stack = Stack()
        embed_g = torch.full((16,), 1.0 + rank)
        lm_head_g = torch.ones(16) * rank
        proj_w_g = torch.ones(4)
        proj_b_g = torch.zeros(4)
        hand_traj_decoder = Head()
        on_grads_ready = None

    m = Model()
    m.hand_traj_decoder.w.grad = torch.full((4,), 2.0 * rank)
    buckets = dp.attach(m, dp.GradBuckets())
    for i in (2, 1, 0):                       # backward order: last layer first
        m.stack.on_layer_grad_ready(i)
    m.on_grads_ready("head")
    m.on_grads_ready("tail")
    buckets.wait()
    ok = True
    for i in range(3):
        ok &= bool(torch.allclose(m.stack.layers[i].flat_g, torch.full((1000,), 3.0 * (i + 1))))   # (1+2)*(i+1)
    ok &= bool(torch.allclose(m.embed_g, torch.full((16,), 3.0)))
    ok &= bool(torch.allclose(m.lm_head_g, torch.full((16,), 1.0)))          # reduced at the 'head' hook
    ok &= bool(torch.allclose(m.stack.norm_g, torch.full((8,), 1.0)))
    ok &= bool(torch.allclose(m.hand_traj_decoder.w.grad, torch.full((4,), 2.0)))
    mean = m.stack.layers[0].flat_g * (1.0 / world)          # what AdamW(grad_scale=1/world) applies
    ok &= bool(torch.allclose(mean, torch.full((1000,), 1.5)))
    ret[rank] = ok
    dist.destroy_process_group()


def test_grad_buckets_allreduce_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
    from hvlm import dp
    b = dp.GradBuckets()
    t = torch.ones(4)
    b.reduce_async(t)
    b.wait()
    assert b.world == 1 and torch.equal(t, torch.ones(4))
