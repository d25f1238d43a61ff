#!/bin/bash
# same-box A/B of the CLIP LayerNorm fold from rocprofv3 kernel stats (round-3 verdict item 4) + the per-shape GEMM table of the step
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fold in 1 0; do
  export HVLM_CLIP_FOLD_LN=$fold
  rm -rf gpurun_out/prof_fold$fold
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fold$fold -o fold$fold -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eval --no-other-configs > gpurun_out/r04b_fold${fold}_bench.json 2>/dev/null
  python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_fold$fold/*.db | head -1)" gpurun_out/r04b_fold${fold}_kernel_stats.txt > /dev/null
  rm -rf gpurun_out/prof_fold$fold
done
unset HVLM_CLIP_FOLD_LN
python bench.py --steps 4 --warmup 2 --gemm-table --no-cpu-baseline --no-eval --no-other-configs > gpurun_out/r04b_bench_table.json 2>/dev/null
cp gpurun_out/gemm_by_shape.json gpurun_out/r04b_gemm_by_shape.json
head -30 gpurun_out/r04b_fold1_kernel_stats.txt
