#!/usr/bin/env python3
"""rocprofv3 CSVs of tools/mfma_pmc.sh -> per-dispatch-group table + gpurun_out/mfma_pmc.json.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the launch's SIMD cycles in which the
matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs; profiles/r02l).
Dispatches are grouped by (kernel, position in tools/mfma_pmc.py's launch order): every group of 3 consecutive launches of one kernel
is one shape."""
import csv, glob, json, sys, collections
d = sys.argv[1]
disp = collections.OrderedDict()
for f in glob.glob(d + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"kernel": r["Kernel_Name"], "c": {}})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
trace = []
for f in glob.glob(d + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        trace.append((int(r["Start_Timestamp"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
trace.sort()
want = lambda k: any(s in k for s in ("gemm_pp", "attn", "hvlm_fa"))
seq_p = [(i, e) for i, e in sorted(disp.items()) if want(e["kernel"])]
seq_t = [(k, us) for _, k, us in trace if want(k)]
LABELS = ["gemm bf16 13872x12288x4096", "gemm bf16 13872x22016x4096", "gemm bf16 13872x4096x11008", "gemm bf16 13872x4096x4096",
          "gemm f16 411200x3072x1024", "gemm f16 411200x1024x1024", "gemm f16 411200x4096x1024", "gemm f16 411200x1024x4096"]
rows, i = [], 0
while i < len(seq_p):
    k = seq_p[i][1]["kernel"]
    j = i
    while j < len(seq_p) and seq_p[j][1]["kernel"] == k and j - i < 3: j += 1
    grp = [e["c"] for _, e in seq_p[i:j]]
    m = {n: sum(g.get(n, 0.0) for g in grp) / len(grp) for n in grp[0]}
    us = [u for kk, u in seq_t[i:j] if kk == k] if len(seq_t) == len(seq_p) else []
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    row = {"kernel": k[:100], "group": len(rows), "launches": j - i, **m,
           "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui) if gui else None,
           "valu_per_mfma": m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"] if m.get("SQ_INSTS_MFMA") else None,
           "avg_us_unprofiled_pass": sum(us) / len(us) if us else None}
    if len(rows) < len(LABELS): row["label"] = LABELS[len(rows)]
    rows.append(row)
    print(f"{row.get('label', ''):30s} {k[:70]:70s} n={j - i} mfma_busy {100 * (row['mfma_busy_frac'] or 0):5.1f} %  VALU/MFMA {row['valu_per_mfma'] or 0:6.2f}  "
          f"avg {row['avg_us_unprofiled_pass'] or 0:9.1f} us")
    i = j
json.dump({"definition": __doc__, "per_group": rows}, open(d + "/../mfma_pmc.json", "w"), indent=1)
