#!/bin/bash
# kernel mix and GPU idle time of ONE fp8 evaluation batch (32 samples, all e4m3 switches): rocprofv3 kernel trace of
# tools/eval_batch.py, the last of five batches
OUT=$GRAFT_REPO_ROOT/gpurun_out/evb
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o evb --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/eval_batch.py > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^batch" $OUT.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
starts = [s for s, e, n in rows if "im2col" in n]
sel = [(s, e, n) for s, e, n in rows if s >= starts[-2]]          # two CLIP chunks per batch: the second-to-last im2col opens the last batch
wall, busy = sel[-1][1] - sel[0][0], sum(e - s for s, e, n in sel)
print(f"last batch: wall {wall / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms, idle {(wall - busy) / 1e6:.1f} ms, {len(sel)} launches")
acc = collections.defaultdict(float)
for s, e, n in sel: acc[n.replace("hvlm::(anonymous namespace)::", "").replace("void ", "")[:64]] += (e - s) / 1e6
for n, v in sorted(acc.items(), key=lambda kv: -kv[1])[:18]: print(f"  {v:8.1f} ms  {n}")
PY
