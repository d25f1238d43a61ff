"""Data-parallel SFT step with the real kernels at world size 2 (reference recipe scripts/finetune.sh:6,20 —
one process per GPU, per-device minibatches, DeepSpeed gradient averaging).  The GPU box has ONE device, so the two ranks
share cuda:0 and exchange over gloo; hooks, buckets, the overlapped per-bucket optimizer and the kernels are the
production ones.  Checks: reduced gradients x 1/2 == mean of the two single-rank gradients (one bf16 rounding of the sum),
both ranks end with identical weights, and those equal a single-process AdamW update on the reduced buckets."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dp_gpu_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(args, **kw):
    return subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], **kw)


def test_dp_world2_real_kernels_shared_device(tmp_path):
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in (0, 1)]
    procs = [_run([r, 2, port, outs[r]]) for r in (0, 1)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    singles = [str(tmp_path / f"s{r}.npz") for r in (0, 1)]
    for r in (0, 1):
        assert _run([r, 1, 0, singles[r], "single"]).wait(timeout=600) == 0
    r0, r1 = np.load(outs[0]), np.load(outs[1])
    s0, s1 = np.load(singles[0]), np.load(singles[1])
    assert abs(float(r0["loss"]) - float(s0["loss"])) < 1e-6 and abs(float(r1["loss"]) - float(s1["loss"])) < 1e-6
    assert int(r0["updated_early"]) >= 3                      # head + both layers were updated from the hooks, not at the end
    gkeys = [k for k in r0.files if k.startswith("g.")]
    assert len(gkeys) > 10
    for k in gkeys:
        want = s0[k].astype(np.float64) + s1[k].astype(np.float64)
        assert np.array_equal(r0[k], r1[k]), k
        # every gradient — the text-embedding rows included: hvlm_embed_grad_segments sums them in a fixed order, no atomics —
        # is deterministic: the reduced bucket = the sum of the two single-rank buckets rounded once to bf16
        tol = 2.0 ** -7 * max(np.abs(want).max(), 1e-30)
        assert np.abs(r0[k] - want).max() <= tol, (k, np.abs(r0[k] - want).max(), tol)
    for k in r0.files:
        if k.startswith("w."):
            assert np.array_equal(r0[k], r1[k]), k
    # single-process AdamW on the reduced buckets reproduces the DP weights bit for bit
    from test_gpu_model import tiny_model
    from hvlm.optim import AdamW
    m, _ = tiny_model(training=True)
    opt = AdamW(m, lr=1e-2)
    for n, _, g, _ in m.trainable_tensors():
        g.copy_(torch.from_numpy(r0["g." + n]).view_as(g))
    for n, p in m.hand_traj_decoder.named_parameters():
        p.grad = torch.from_numpy(r0["g.head." + n]).cuda()
    opt.step(grad_scale=0.5)
    torch.cuda.synchronize()
    for n, w, _, _ in m.trainable_tensors():
        assert np.array_equal(w.float().cpu().numpy(), r0["w." + n]), n
    for n, p in m.hand_traj_decoder.named_parameters():
        assert np.array_equal(p.detach().float().cpu().numpy(), r0["w.head." + n]), n


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside a launcher starts two fresh rank processes itself (never re-executing a process
    that touched the GPU) and prints rank 0's single JSON line; with fewer GPUs than ranks it exits 3 with a clear message
    unless HVLM_SHARE_DEVICE=1 (both ranks on device 0 over gloo, functional check)."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--model", "tiny", "--batch", "2", "--text-len", "64",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-eval"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HVLM_SHARE_DEVICE="0"))
        assert r.returncode == 3 and "needs 2 visible GPUs" in r.stderr
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HVLM_SHARE_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 4
    assert np.isfinite(res["value"]) and res["value"] > 0
    _check_comm(res, 2)


def _check_comm(res, world, layers=2):
    """The `comm` object of an N > 1 (or forced-distributed) bench line: one collective per decoder-layer bucket + lm_head + the
    tail's tensors, the bytes handed to them, and the exposed tail in HIP-event milliseconds."""
    c = res["comm"]
    assert c["world"] == world and c["collectives_per_step"] >= layers + 1 + 4
    assert c["bytes_allreduced_per_step"] > 2 * 32101 * 128 * 2          # at least embed_tokens + lm_head gradients (bf16)
    assert c["exposed_ms"] is not None and c["exposed_ms"] > 0 and c["exposed_ms"] == res["step_tail_ms"]
    assert abs(c["ring_bytes_per_gpu"] - 2.0 * (world - 1) / world * c["bytes_allreduced_per_step"]) < 1.0
    # straggler skew and the communication library's identity ride on the line too (round-3 verdict item 6)
    sk = c["rank_skew"]
    assert len(sk["per_rank_step_ms_mean_min_max"]) == world and all(lo <= mean <= hi for mean, lo, hi in sk["per_rank_step_ms_mean_min_max"])
    assert 0 <= sk["slowest_rank"] < world and sk["mean_step_ms_spread"] >= 0
    assert "env" in c["library"] and "hip" in c["library"]
    assert c["embed_rows_allreduced"] is None                              # dense embedding all-reduce unless HVLM_DP_EMBED_ROWS=1


def test_bench_world8_sharing_one_device(tmp_path):
    """`bench.py --gpus 8` end to end on this 1-GPU box: eight rank processes on device 0 over gloo (HVLM_SHARE_DEVICE=1) — the
    spawn path, the rendezvous, 8-way bucketed all-reduces from the backward hooks, the overlapped optimizer, the barrier and the
    MAX over ranks of the timing, rank 0's single JSON line with global batch 8 x 2 and the `comm` object."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--model", "tiny", "--batch", "2", "--text-len", "32",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-eval"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, HVLM_SHARE_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["config"]["parallelism"] == "dp8" and res["config"]["global_batch"] == 16
    assert np.isfinite(res["value"]) and res["value"] > 0 and np.isfinite(res["loss"])
    _check_comm(res, 8)


def test_bench_under_torch_distributed_run(tmp_path):
    """The launch line the driver uses for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — on this 1-GPU box with both ranks on device 0 over
    gloo (HVLM_SHARE_DEVICE=1, HVLM_DIST_BACKEND=gloo): rank 0 prints the single JSON line, value = both ranks' samples."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "bench.py"), "--gpus", "2", "--model", "tiny", "--batch", "2",
           "--text-len", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-eval"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HVLM_SHARE_DEVICE="1", HVLM_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 4 and res["value"] > 0


def test_bench_rccl_backend_world1(tmp_path):
    """The REAL collective backend ("nccl" = RCCL) on this 1-GPU box: HVLM_FORCE_DIST=1 makes the engine take the distributed
    path at world size 1 — process-group init over 127.0.0.1, one asynchronous RCCL all-reduce per gradient bucket on RCCL's
    stream, the overlapped optimizer ordered behind the work handles, the barrier and the MAX reduction of the timing — so
    everything except the link traffic of N > 1 is exercised with the library the 8-GPU run uses.  Same loss as the
    non-distributed run (a SUM over one rank with grad_scale 1 is the identity)."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--model", "tiny", "--batch", "2", "--text-len", "64",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-eval"]
    res = {}
    for tag, extra in (("plain", {}), ("rccl", dict(HVLM_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                                                     MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port()))):
        env = {k: v for k, v in os.environ.items() if k not in ("HVLM_DIST_BACKEND", "HVLM_SHARE_DEVICE")}
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, **extra))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, r.stdout[-1000:]
        res[tag] = json.loads(lines[0])
    assert res["rccl"]["n_gpus"] == 1 and res["rccl"]["value"] > 0
    _check_comm(res["rccl"], 1)
    assert "comm" not in res["plain"] and res["plain"]["step_tail_ms"] > 0
    for r in res.values():                      # per-step diagnostic: one entry per timed step, consistent with the wall-clock mean
        assert len(r["step_ms"]) == 3 and all(v > 0 for v in r["step_ms"]) and r["device_allocs_in_timed_region"] >= 0
        assert abs(sum(r["step_ms"]) / 3 - r["ms_per_step"]) <= 0.25 * r["ms_per_step"] + 1.0
    assert res["rccl"]["loss"] == res["plain"]["loss"], (res["rccl"]["loss"], res["plain"]["loss"])
