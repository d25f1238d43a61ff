#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 300 python3 tools/attn_clip_check.py > gpurun_out/r05l_clip_check.log 2>&1; tail -12 gpurun_out/r05l_clip_check.log
timeout 300 python3 tools/attn_clip_prof.py > gpurun_out/r05l_clip_prof.log 2>&1; cat gpurun_out/r05l_clip_prof.log
python -m pytest tests/test_gpu_r5.py -q -k clip 2>&1 | tail -3
timeout 900 python3 tools/whatif_r5.py > gpurun_out/r05l_whatif.txt 2>gpurun_out/r05l_whatif.err; cat gpurun_out/r05l_whatif.txt; tail -3 gpurun_out/r05l_whatif.err
# in-situ cycles of the assembly kernels (the C++ half is gpurun_out/r05k_insitu_pmc.txt); a hung profiler must not eat the budget
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT}"
OUT=gpurun_out/r05l_pmc; rm -rf $OUT; mkdir -p $OUT
HVLM_ATTN_FWD_ASM=1 HVLM_ATTN_CLIP_ASM=1 timeout 700 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/asm1 -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-other-configs --no-h2d --no-gemm-events > $OUT/log.txt 2>&1
{ echo "== asm=1  GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; python3 tools/insitu_pmc.py $OUT/asm1; } > gpurun_out/r05l_insitu_pmc_asm1.txt 2>&1
rm -rf $OUT; cat gpurun_out/r05l_insitu_pmc_asm1.txt
