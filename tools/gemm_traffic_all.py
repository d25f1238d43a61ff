#!/usr/bin/env python3
"""HBM-side (L2-miss / fabric) traffic of the GEMM kernel per launch for EVERY GEMM shape of a bench.py run, from PMC
counters, in two rocprofv3 passes (FETCH_SIZE alone, WRITE_SIZE alone — MI355X_MICROARCH.md: separate --pmc passes; gfx950:
FETCH_SIZE tallies 128-B requests at 64 B -> x2 for wide coalesced reads; units KB).

    python3 tools/gemm_traffic_all.py drive [gpurun_out/gemm_by_shape.json] [out.json]     (on the GPU box)

`drive` runs `rocprofv3 --kernel-trace --pmc <CTR> -- python3 tools/gemm_traffic_all.py run <shapes.json>` once per counter:
ONE process walks all shapes (3 calls of ops.gemm_nt each, operands = views of two buffers filled once), so the cost is two
profiler start-ups instead of two per shape (tools/gemm_traffic.py, round 2: ~12 GPU-minutes for 21 shapes).  The `run` side
writes the number of GEMM-kernel launches each call made (a tail-split call is two) to <shapes>.launches.json; `drive` walks
the counter CSV in dispatch order and hands consecutive gemm_pp dispatches to the shapes accordingly.
Per shape it reports bytes per CALL (all launches of a call summed) and the launches per call, like the round-2 table.

    python3 tools/gemm_traffic_all.py drive-fp8 [out.json]

is the same for the e4m3 GEMMs of BASELINE config 5 (bench.py `eval.batched_fp8`): the four decoder-prefill shapes at
M = 32 x 867 rows (q|k|v, o_proj, gate|up, down_proj; per-token x per-channel scales applied in the epilogue) through
ops.gemm_fp8_nt_scaled -> gemm_pp4_fp8_kernel; algorithmic bytes = 1 B per operand element + 2 B per output element + the scales.
The table is committed as profiles/r*_gemm_traffic_fp8.json and read by bench.py for `eval.batched_fp8.roofline.traffic`."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = 3


FP8_SHAPES = [dict(M=32 * 867, N=12288, K=4096, launches=32), dict(M=32 * 867, N=4096, K=4096, launches=32),
              dict(M=32 * 867, N=22016, K=4096, launches=32), dict(M=32 * 867, N=4096, K=11008, launches=32)]


def run(shapes_path, fp8=False):
    sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
    import torch
    from hvlm import ops
    shapes = [r for r in json.load(open(shapes_path)) if r["K"] % 64 == 0]
    if fp8:
        return run_fp8(shapes, shapes_path, torch, ops)
    BF = torch.bfloat16
    na = max(r["M"] * r["K"] for r in shapes)
    nb = max(r["N"] * r["K"] for r in shapes)
    nc = max(r["M"] * r["N"] for r in shapes)
    abuf = torch.rand(na, device="cuda").sub_(0.5).to(BF)
    bbuf = torch.rand(nb, device="cuda").sub_(0.5).to(BF)
    cbuf = torch.empty(nc, dtype=BF, device="cuda")
    launches = []
    cur = [0]

    def hook(M, N, K, begin, n_kernels=1):
        if not begin:
            cur[0] += n_kernels
    ops.set_gemm_hook(hook)
    for r in shapes:
        M, N, K = r["M"], r["N"], r["K"]
        a, b, out = abuf[: M * K].view(M, K), bbuf[: N * K].view(N, K), cbuf[: M * N].view(M, N)
        cur[0] = 0
        for _ in range(CALLS):
            ops.gemm_nt(a, b, out=out)
        torch.cuda.synchronize()
        launches.append(dict(M=M, N=N, K=K, kernels=cur[0]))
    json.dump(launches, open(shapes_path + ".launches.json", "w"))


def run_fp8(shapes, shapes_path, torch, ops):
    BF = torch.bfloat16
    launches = []
    cur = [0]

    def hook(M, N, K, begin, n_kernels=1):
        if not begin:
            cur[0] += n_kernels
    ops.set_gemm_hook(hook)
    for r in shapes:
        M, N, K = r["M"], r["N"], r["K"]
        a8, ia = ops.quantize_fp8_rowwise(torch.rand(M, K, device="cuda").sub_(0.5).to(BF))
        b8, ib = ops.quantize_fp8_rowwise(torch.rand(N, K, device="cuda").sub_(0.5).to(BF))
        out = torch.empty(M, N, dtype=BF, device="cuda")
        cur[0] = 0
        for _ in range(CALLS):
            ops.gemm_fp8_nt_scaled(a8, b8, ia, ib, out=out)
        torch.cuda.synchronize()
        launches.append(dict(M=M, N=N, K=K, kernels=cur[0]))
        del a8, b8, out
    json.dump(launches, open(shapes_path + ".launches.json", "w"))


def drive(shapes_path, out_path, fp8=False):
    by_ctr = {}
    kname = "gemm_pp4_fp8" if fp8 else "gemm_pp"
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_all_{ctr}")
        subprocess.run(["rm", "-rf", d])
        subprocess.run(["timeout", "900", "rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
                        "python3", os.path.abspath(__file__), "run-fp8" if fp8 else "run", os.path.abspath(shapes_path)],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        rows = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if kname in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        rows.sort()
        by_ctr[ctr] = [v for _, v in rows]
        subprocess.run(["rm", "-rf", d])
    launches = json.load(open(shapes_path + ".launches.json"))
    table = {(r["M"], r["N"], r["K"]): r["launches"] for r in json.load(open(shapes_path))}
    need = sum(r["kernels"] for r in launches)
    for ctr, vals in by_ctr.items():
        if len(vals) != need:
            raise SystemExit(f"{ctr}: {len(vals)} gemm_pp dispatches in the counter CSV, expected {need}")
    out, pos = [], 0
    for r in launches:
        n = r["kernels"]
        f = sum(by_ctr["FETCH_SIZE"][pos:pos + n]) / CALLS
        w = sum(by_ctr["WRITE_SIZE"][pos:pos + n]) / CALLS
        pos += n
        M, N, K = r["M"], r["N"], r["K"]
        out.append(dict(M=M, N=N, K=K, launches=table[(M, N, K)], kernels_per_call=n / CALLS, fetch_bytes=f * 1024 * 2,
                        write_bytes=w * 1024,
                        algorithmic_bytes=(1.0 * (M * K + N * K) + 2.0 * M * N + 4.0 * (M + N)) if fp8 else 2.0 * (M * K + N * K + M * N)))
        print(out[-1], flush=True)
    n = sum(o["launches"] for o in out)
    res = dict(per_shape=out,
               mean_traffic_bytes_per_call=sum(o["launches"] * (o["fetch_bytes"] + o["write_bytes"]) for o in out) / n,
               mean_algorithmic_bytes_per_call=sum(o["launches"] * o["algorithmic_bytes"] for o in out) / n,
               calls_covered=n, calls_total=sum(table.values()),
               method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), one process over all shapes "
                      "(tools/gemm_traffic_all.py), per ops.gemm_nt call (all launches of the GEMM kernel in the call summed), "
                      "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), KB -> B, call-weighted")
    json.dump(res, open(out_path, "w"), indent=1)
    print({k: v for k, v in res.items() if k != "per_shape"})


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "drive"
    shapes = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "gemm_by_shape.json")
    if mode == "run":
        run(shapes)
    elif mode == "run-fp8":
        run(shapes, fp8=True)
    elif mode == "drive-fp8":
        sp = os.path.join(ROOT, "gpurun_out", "gemm_fp8_shapes.json")
        os.makedirs(os.path.dirname(sp), exist_ok=True)
        json.dump(FP8_SHAPES, open(sp, "w"))
        drive(sp, sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "gemm_traffic_fp8.json"), fp8=True)
    else:
        drive(shapes, sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "gemm_traffic.json"))
