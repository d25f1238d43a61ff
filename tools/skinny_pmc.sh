#!/bin/bash
# PMC passes over tools/skinny_gemm_bench.py: memory-side fetch bytes, L2 hit / miss, TCP -> L2 requests (one pass per group)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/skinny_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python3 tools/skinny_gemm_bench.py > $OUT/log$i.txt 2>&1
  tail -2 $OUT/log$i.txt
done
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "skinny" in r["Kernel_Name"] or "gemm_pp" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][29:52], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    print(f"{k:24s} grid={g:>8s} {c:30s} n={len(v):3d} mean={sum(v)/len(v):.4e}")
PY
cat $OUT/summary.txt
