#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters collected INSIDE the bench step (counter_collection.csv joined with the kernel trace
for durations): kernel, dispatches, mean duration, each counter's mean, and — when GRBM_GUI_ACTIVE is there — the shader clock the
kernel ran at (GUI_ACTIVE sums over the 8 XCDs: cycles = value / 8).

    python3 tools/insitu_pmc.py <rocprofv3 output dir> [substring ...]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
want = sys.argv[2:] or ["hvlm_fa_", "attn_fwd", "attn_bwd", "gemm_pp4m"]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
seen = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(w in k for w in want):
            continue
        k = k.replace("hvlm::(anonymous namespace)::", "").replace("void ", "")[:44]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        did = r["Dispatch_Id"]
        if did not in seen[k]:
            seen[k].add(did)
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                acc[k]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
            elif did in dur:
                acc[k]["_us"].append(dur[did])
for k, c in sorted(acc.items()):
    us = c.pop("_us", [])
    mean_us = sum(us) / max(1, len(us))
    line = f"{k:44s} n={len(us):4d}  {mean_us:8.1f} us"
    for name, v in sorted(c.items()):
        m = sum(v) / len(v)
        line += f"  {name} {m:.4e}"
        if name == "GRBM_GUI_ACTIVE" and mean_us > 0:
            line += f" (clock {m / 8 / mean_us * 1e-3:.2f} GHz)"
    print(line)
