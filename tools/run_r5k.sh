#!/bin/bash
# In-situ counters of the attention kernels (and the GEMMs beside them) inside the bench step, C++ vs assembly: cycles, MFMA-busy cycles,
# fabric bytes — does the same wall time hide different cycle counts (clock) or different memory traffic?
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tag=${1:-r05k}; OUT=gpurun_out/$tag; rm -rf $OUT; mkdir -p $OUT
for asm in 0 1; do
  for c in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE WRITE_SIZE"; do
    t=$(echo $c | tr ' ' '+')
    HVLM_ATTN_FWD_ASM=$asm HVLM_ATTN_CLIP_ASM=$asm rocprofv3 --pmc $c --kernel-trace -d $OUT/asm${asm}_$t -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-other-configs --no-h2d --no-gemm-events > $OUT/log_asm${asm}_$t.txt 2>&1
    echo "== asm=$asm  $c"; python3 tools/insitu_pmc.py $OUT/asm${asm}_$t
  done
done > gpurun_out/${tag}_insitu_pmc.txt 2>&1
du -sh $OUT | cut -f1; find $OUT -name "*.csv" -size +8M -delete; rm -rf $OUT/*/*/*.db
cat gpurun_out/${tag}_insitu_pmc.txt
