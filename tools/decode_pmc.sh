#!/bin/bash
# memory-side fetch bytes (FETCH_SIZE, KB; x2 on gfx950, MI355X_MICROARCH.md) of the decode-step kernels: tools/decode_batched_bench.py
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/decode_pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python3 tools/decode_batched_bench.py > $OUT/log1.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_decode_batched" in n or "gemm_skinny" in n:
            acc[(n.replace("hvlm::(anonymous namespace)::", "").replace("void ", "")[:40], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    print(f"{k:42s} grid={g:>8s} {c:12s} n={len(v):4d} mean={sum(v) / len(v):.4e} KB  -> x2 = {2 * sum(v) / len(v) / 1e3:.1f} MB per launch")
PY
