#!/bin/bash
# kernel stats of the fp8 batch-evaluation leg (bench.py --fp8-eval): which kernels the 32-sample batch spends its time in
OUT=$GRAFT_REPO_ROOT/gpurun_out/eval_${1:-a}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o ev --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --fp8-eval --steps 2 --warmup 1 > $OUT.json 2> $OUT.err
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms")
for r in rows[:28]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"]) / 1e6:8.1f} avg_us {float(r["AverageNs"]) / 1e3:9.1f}')
PY
