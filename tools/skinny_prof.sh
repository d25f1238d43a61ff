#!/bin/bash
# per-kernel durations of tools/skinny_gemm_bench.py (kernel trace, grouped by kernel and grid)
OUT=$GRAFT_REPO_ROOT/gpurun_out/sk_${1:-a}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o sk --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/skinny_gemm_bench.py > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:48], r.get("Grid_Size_X"), r.get("Grid_Size_Y"))
    d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    if len(v) < 20: continue
    v = sorted(v); print(k, len(v), "median_ns", v[len(v) // 2], "min_ns", v[0])
PY
