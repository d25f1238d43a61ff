#!/usr/bin/env python3
"""Tile-cycle timeline of the persistent ping-pong GEMM from in-kernel s_memtime stamps (library built by tools/gemm_tile_timeline.sh,
selected with HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_prof.so).  For each shape: one warm launch, clear, one launch, then per
wave and tile (XCD 0's first 8 CUs, tiles 1..31 of each workgroup) the durations  K loop | bookkeeping before the epilogue | epilogue |
bookkeeping after | first K-tile, in stamp ticks, and in microseconds by the measured tick rate (launch wall time / stamp span)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import numpy as np
import torch
from hvlm import ops, _lib

lib = ctypes.CDLL(os.environ["HVLM_LIB"])
lib.hvlm_debug_gemm_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
F16, BF = torch.float16, torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)


def run(name, M, N, K, dtype, kind):
    rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(dtype)  # noqa: E731
    a, w = rn(M, K), rn(N, K, sc=K ** -0.5)
    out = torch.empty(M, N, dtype=dtype, device="cuda")
    bias = rn(N, sc=0.1)
    if kind == "plain":
        fn = (lambda: ops.gemm_f16_nt(a, w, out=out)) if dtype == F16 else (lambda: ops.gemm_nt(a, w, out=out))
    elif kind == "ln_gelu":
        st = torch.stack([a.float().mean(1), torch.rsqrt(a.float().var(1, unbiased=False) + 1e-5)], 1).contiguous()
        c1 = w.float().sum(1).contiguous()
        fn = lambda: ops.gemm_f16_nt_ln(a, w, bias=bias, act=ops.ACT_QUICK_GELU, out=out, ln=(st, c1))
    elif kind == "bias":
        fn = lambda: ops.gemm_f16_nt(a, w, bias=bias, out=out)
    elif kind in ("res", "res_nostats", "res_only"):
        x = rn(M, N)
        part = torch.empty((M, ops.ln_stats_parts(N), 2), dtype=torch.float32, device="cuda")
        if kind == "res":
            fn = lambda: ops.gemm_f16_nt_ln(a, w, bias=bias, residual=x, out=out, stats_out=part)
        elif kind == "res_nostats":
            fn = lambda: ops.gemm_f16_nt(a, w, bias=bias, residual=x, out=out)
        else:
            fn = (lambda: ops.gemm_f16_nt(a, w, residual=x, out=out)) if dtype == F16 else (lambda: ops.gemm_nt(a, w, residual=x, out=out))
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    assert lib.hvlm_debug_gemm_prof(None, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = np.zeros((8, 8, 32, 12), dtype=np.uint64)
    assert lib.hvlm_debug_gemm_prof(buf.ctypes.data, 0) == 0
    t = buf.astype(np.int64)
    valid = t[:, :, :, 0] > 0
    nt = int(valid[0, 0].sum())
    if nt < 3:
        t = t[valid.any((1, 2))]                # few tiles: only the workgroups that ran
    span = (t[..., 3][valid].max() - t[..., 0][valid].min())
    # stamp ticks per microsecond from the kernel's own span of the sampled workgroups vs the launch duration (close to each other
    # when the sampled workgroups run from the first to the last round)
    tiles_per_wg = (M + 255) // 256 * (N // 256) / 256.0
    tick_us = (float(np.mean(t[:, :, 1:nt - 1, 0] - t[:, :, :nt - 2, 0])) * tiles_per_wg / (ms * 1e3)) if nt >= 3 else 1500.0   # (tile period x tiles per workgroup) / launch time
    its = slice(1, max(2, nt - 1)) if nt >= 3 else slice(0, max(1, nt))
    seg = {"k_loop": t[..., 1] - t[..., 0], "pre_epi": t[..., 2] - t[..., 1], "epilogue": t[..., 3] - t[..., 2],
           "post_epi": t[..., 4] - t[..., 3], "first_ktile": t[..., 5] - t[..., 0]}
    if kind.startswith("res"):
        seg.update({"res_head": t[..., 6] - t[..., 2], "res_i0": t[..., 7] - t[..., 6], "res_i1": t[..., 8] - t[..., 7],
                    "res_i2": t[..., 9] - t[..., 8], "res_i3": t[..., 10] - t[..., 9]})
    period = t[:, :, 1:, 0] - t[:, :, :-1, 0]
    print(f"== {name} {M}x{N}x{K} {kind}: {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s, {ms * 1e3:.0f} us, {tiles_per_wg:.1f} tiles per workgroup, "
          f"{nt} stamped; ~{tick_us:.1f} ticks/us")
    per = period[:, :, its.start:its.stop - 1] if nt >= 3 else np.zeros((1,))
    print(f"   tile period  mean {per.mean():8.1f} ticks = {per.mean() / tick_us:6.2f} us   (min {per.min()}, max {per.max()})")
    for k, v in seg.items():
        x = v[:, :, its]
        g0, g1 = x[:, :4], x[:, 4:]
        print(f"   {k:12s} mean {x.mean():8.1f} ticks = {x.mean() / tick_us:6.2f} us   group0 {g0.mean():8.1f}  group1 {g1.mean():8.1f}   "
              f"min {x.min()} max {x.max()}")
    # skew between the waves of one workgroup at the epilogue entry / exit
    e_in = t[:, :, its, 2]; e_out = t[:, :, its, 3]
    print(f"   epilogue entry spread over the 8 waves {np.mean(e_in.max(1) - e_in.min(1)):.1f} ticks, exit spread {np.mean(e_out.max(1) - e_out.min(1)):.1f}; "
          f"first wave in -> last wave out {np.mean(e_out.max(1) - e_in.min(1)):.1f} ticks = {np.mean(e_out.max(1) - e_in.min(1)) / tick_us:.2f} us")
    return t


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 411200
    ts = {}
    ts["fc1_plain"] = run("fc1", M, 4096, 1024, F16, "plain")
    ts["fc1_bias"] = run("fc1", M, 4096, 1024, F16, "bias")
    ts["fc1"] = run("fc1", M, 4096, 1024, F16, "ln_gelu")
    ts["out"] = run("out", M, 1024, 1024, F16, "res")
    ts["out_nostats"] = run("out", M, 1024, 1024, F16, "res_nostats")
    ts["out_resonly"] = run("out", M, 1024, 1024, F16, "res_only")
    ts["fc2_plain"] = run("fc2", M, 1024, 4096, F16, "plain")
    ts["dec_gu"] = run("gate|up", 13872, 22016, 4096, BF, "plain")
    ts["dec_o"] = run("o_proj", 13872, 4096, 4096, BF, "res_only")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "gemm_tile_timeline.npz"), **ts)
