#!/bin/bash
# per-kernel durations of one lock-step decode step at batch 32 (tools/decode_batched_bench.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/dec_${1:-a}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o dec --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/decode_batched_bench.py > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
grep "decode step" $OUT.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_ns {float(r["AverageNs"]):9.0f} pct {r["Percentage"]}')
PY
