#!/usr/bin/env python3
"""HBM-side traffic of the GEMM kernel per launch, from PMC counters (GPU box).

For every distinct GEMM shape of the SFT step (gpurun_out/gemm_by_shape.json from `bench.py --gemm-table`) runs
`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (each ALONE in its pass, as MI355X_MICROARCH.md prescribes) on
tools/gemm_one.py, applies the gfx950 correction (FETCH_SIZE tallies 128-B requests at 64 B: x2 for wide coalesced
reads; units KB) and writes the launch-weighted mean to gpurun_out/gemm_traffic.json.  A whole-step PMC pass is not
possible on this pool (the ~3000-dispatch step under --pmc exceeds the time limit), hence per shape in isolation:
L2-miss traffic of a kernel whose operands were just produced by the previous kernel can only be lower in situ."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = json.load(open(os.path.join(ROOT, "gpurun_out", "gemm_by_shape.json")))
out = []
for r in shapes:
    M, N, K = r["M"], r["N"], r["K"]
    if K % 64 or r["launches"] < 3:
        continue
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_{ctr}_{M}x{N}x{K}")
        subprocess.run(["timeout", "150", "rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--",
                        "python3", os.path.join(ROOT, "tools", "gemm_one.py"), str(M), str(N), str(K), "3"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        v = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "gemm_pp" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    v.append(float(row["Counter_Value"]))
        # per CALL of ops.gemm_nt (3 calls): a tail-split call is two launches of the GEMM kernel, both are summed
        vals[ctr] = sum(v) / 3.0 if v else None
        vals["kernels_per_call"] = len(v) / 3.0
        subprocess.run(["rm", "-rf", d])
    if vals["FETCH_SIZE"] is None or vals["WRITE_SIZE"] is None:
        continue
    fetch_b = vals["FETCH_SIZE"] * 1024 * 2      # KB -> B, gfx950: 128-B requests counted as 64 B
    write_b = vals["WRITE_SIZE"] * 1024
    algo = 2.0 * (M * K + N * K + M * N)
    out.append(dict(M=M, N=N, K=K, launches=r["launches"], kernels_per_call=vals["kernels_per_call"], fetch_bytes=fetch_b, write_bytes=write_b,
                    algorithmic_bytes=algo))
    print(out[-1], flush=True)
n = sum(o["launches"] for o in out)
res = dict(per_shape=out,
           mean_traffic_bytes_per_launch=sum(o["launches"] * (o["fetch_bytes"] + o["write_bytes"]) for o in out) / n,
           mean_algorithmic_bytes_per_launch=sum(o["launches"] * o["algorithmic_bytes"] for o in out) / n,
           launches_covered=n, launches_total=sum(r["launches"] for r in shapes),
           method="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/gemm_one.py per shape, per ops.gemm_nt call "
                  "(all launches of the GEMM kernel in the call summed), FETCH_SIZE x2 (gfx950), call-weighted")
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_traffic.json"), "w"), indent=1)
print({k: v for k, v in res.items() if k != "per_shape"})
