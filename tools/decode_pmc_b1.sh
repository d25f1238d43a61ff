#!/bin/bash
# FETCH_SIZE (KB; x2 on gfx950 for wide reads, MI355X_MICROARCH.md) of the batch-1 decode step's kernels, per launch and summed per token
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/decode_pmc_b1; rm -rf $OUT; mkdir -p $OUT
N=3
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python3 tools/decode_pmc_b1.py $N > $OUT/log1.txt 2>&1
python3 - $OUT $N <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("gemv" in n or "attn_decode" in n) and r["Counter_Name"] == "FETCH_SIZE":
            acc[n.replace("hvlm::(anonymous namespace)::", "").replace("void ", "")[:34]].append(float(r["Counter_Value"]))
steps = int(sys.argv[2]) + 1
tot = 0.0
rows = {}
for k, v in sorted(acc.items()):
    per = 2 * sum(v) / len(v) * 1024
    tot += 2 * sum(v) * 1024 / steps
    rows[k] = dict(launches_per_token=len(v) / steps, fetch_mb_per_launch=per / 1e6)
    print(f"{k:36s} launches/token {len(v) / steps:6.1f}  FETCH_SIZE x2 = {per / 1e6:8.2f} MB per launch")
alg = 13.41e9
print(f"fetched per token: {tot / 1e9:.3f} GB  (algorithmic: 12.95 GB of layer weights + 0.45 GB of K / V rows = 13.41 GB)  ratio {tot / alg:.3f}")
json.dump(dict(fetch_bytes_per_token=tot, algorithmic_bytes_per_token=alg, per_kernel=rows,
               method="rocprofv3 --pmc FETCH_SIZE (alone in its pass) on tools/decode_pmc_b1.py, x2 (gfx950), KB -> bytes, mean over 4 eager steps; "
                      "WRITE_SIZE not collected (the step writes ~100 KB per token)"), open("gpurun_out/decode_traffic_b1.json", "w"), indent=1)
PY
rm -rf $OUT/p1
