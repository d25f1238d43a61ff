#!/bin/bash
# per-kernel durations + GPU-busy share of the batch-1 evaluation sample (tools/eval_one.py): gpurun_out/<tag>_eval_kernel_stats.txt
tag=${1:-ev}
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 tools/eval_one.py > gpurun_out/${tag}_eval_one.log 2>&1
python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_$tag/*.db | head -1)" gpurun_out/${tag}_eval_kernel_stats.txt > /dev/null
rm -rf gpurun_out/prof_$tag
grep "ms per sample" gpurun_out/${tag}_eval_one.log
head -45 gpurun_out/${tag}_eval_kernel_stats.txt | cut -c1-200
