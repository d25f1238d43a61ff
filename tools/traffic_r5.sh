#!/bin/bash
# Round-5 fabric-traffic measurement: (1) which L2-memory-side counters this rocprofv3 knows (is there an HBM-side one
# that separates MALL hits from DRAM reads?), (2) FETCH_SIZE / DRAM-side read requests / time of the decoder's widest GEMM
# (13872 x 22016 x 4096) for gemm_group_m in {2, 4, 8}: the per-shape XCD block experiment.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/traffic_r5; rm -rf $OUT; mkdir -p $OUT
timeout 60 rocprofv3 -L > $OUT/counters_all.txt 2>&1
grep -i -o -E "\b(TCC_EA0?_[A-Z0-9_]+|TCC_[A-Z0-9_]*DRAM[A-Z0-9_]*|TCC_[A-Z0-9_]*MALL[A-Z0-9_]*|[A-Z0-9_]*HBM[A-Z0-9_]*|FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum)\b" $OUT/counters_all.txt | sort -u > $OUT/counters_memside.txt
echo "memory-side counters:"; cat $OUT/counters_memside.txt | tr '\n' ' '; echo
for g in 4 2 8; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '+')
    HVLM_GEMM_GROUP_M=$g timeout 150 rocprofv3 --pmc $c --kernel-trace -d $OUT/g${g}_$tag -o p --output-format csv -- python3 tools/gemm_one.py 13872 22016 4096 3 > $OUT/log_g${g}_$tag.txt 2>&1 || echo "   ($c: not collected)"
  done
  echo "== gemm_group_m=$g"; python3 tools/pmc_sum.py $OUT/g${g}_FETCH_SIZE 2>/dev/null; for d in $OUT/g${g}_*; do python3 tools/pmc_sum.py $d 2>/dev/null; done | sort -u
done
python3 tools/gemm_knobs.py "HVLM_GEMM_GROUP_M=4" "HVLM_GEMM_GROUP_M=2" "HVLM_GEMM_GROUP_M=8" 2>&1 | tail -4
