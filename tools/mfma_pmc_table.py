#!/usr/bin/env python3
"""rocprofv3 CSVs of tools/mfma_pmc.sh -> per-group table + gpurun_out/mfma_pmc.json.
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the launches' SIMD cycles in which the
matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs; profiles/r02l).
tools/mfma_pmc.py puts a marker dispatch (a torch fill) in front of every group; a group = the MFMA kernels between two markers, summed
per kernel name (a GEMM call whose tail rows run as split-K is two kernels; the attention backward is dQ + fused dK/dV)."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
def load(sub, need_counters):
    rows = collections.OrderedDict()
    pat = "/**/*counter_collection.csv" if need_counters else "/**/*kernel_trace.csv"
    for f in glob.glob(d + "/" + sub + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            e = rows.setdefault(k, {"kernel": r["Kernel_Name"], "c": {}})
            if need_counters: e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
            else: e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return [rows[k] for k in sorted(rows)]
labels = json.load(open(os.path.join(d, "..", "mfma_pmc_labels.json")))
want = lambda k: any(s in k for s in ("gemm_pp", "attn", "hvlm_fa"))
def groups(seq):
    out, cur = [], None
    for e in seq:
        if "FillFunctor" in e["kernel"] and cur is not None or ("FillFunctor" in e["kernel"]):
            cur = []; out.append(cur)
        elif cur is not None and want(e["kernel"]):
            cur.append(e)
    return out
gp, gt = groups(load("p1", True)), groups(load("t", False))
rows = []
for gi, grp in enumerate(gp):
    label = labels[gi] if gi < len(labels) else f"group {gi}"
    per = collections.OrderedDict()
    for e in grp: per.setdefault(e["kernel"], []).append(e["c"])
    tper = collections.OrderedDict()
    if gi < len(gt):
        for e in gt[gi]: tper.setdefault(e["kernel"], []).append(e["us"])
    for k, cs in per.items():
        m = {n: sum(c.get(n, 0.0) for c in cs) / len(cs) for n in cs[0]}
        gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        us = tper.get(k)
        row = {"group": label, "kernel": k[:100], "launches": len(cs), **m,
               "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui) if gui else None,
               "valu_per_mfma": m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"] if m.get("SQ_INSTS_MFMA") else None,
               "avg_us_unprofiled_pass": sum(us) / len(us) if us else None}
        rows.append(row)
        print(f"{label:34s} {k[:64]:64s} n={len(cs)} mfma_busy {100 * (row['mfma_busy_frac'] or 0):5.1f} %  VALU/MFMA {row['valu_per_mfma'] or 0:6.2f}  "
              f"avg {row['avg_us_unprofiled_pass'] or 0:9.1f} us")
json.dump({"definition": __doc__, "per_group": rows}, open(d + "/../mfma_pmc.json", "w"), indent=1)
