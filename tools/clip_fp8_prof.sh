#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/clip8_${1:-a}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o c8 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/clip_fp8_bench.py > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
grep "CLIP tower" $OUT.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} total_ms {float(r["TotalDurationNs"])/1e6:8.1f} avg_us {float(r["AverageNs"])/1e3:9.1f}')
PY
