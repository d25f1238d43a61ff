#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the e4m3 GEMM on config 5's q|k|v prefill shape for gemm_group_m in {1, 2, 4, 8}
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/fp8_traffic; rm -rf $OUT; mkdir -p $OUT
for g in 4 1 2 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    HVLM_GEMM_GROUP_M=$g timeout 150 rocprofv3 --pmc $c --kernel-trace -d $OUT/g${g}_$c -o p --output-format csv -- python3 tools/fp8_one.py 27744 12288 4096 3 > $OUT/log_g${g}_$c.txt 2>&1 || echo "   ($c: not collected)"
  done
  echo "== gemm_group_m=$g (27744 x 12288 x 4096 e4m3; FETCH_SIZE in KB, x2 on gfx950 for wide reads; algorithmic 0.16 + 0.05 GB in, 0.68 GB out)"
  for d in $OUT/g${g}_*; do python3 tools/pmc_sum.py $d 2>/dev/null; done | sort -u
done
rm -rf $OUT/g*_*
