#!/bin/bash
# per-kernel durations + kernel-to-kernel gaps of the batch-1 KV-cached decode step (tools/decode_bench.py: 7B stack, cache length ~880)
# usage: bash tools/decode_trace.sh <tag>      -> gpurun_out/<tag>_decode_kernel_stats.txt
tag=${1:-dec}
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 tools/decode_bench.py > gpurun_out/${tag}_decode_bench.log 2>&1
python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_$tag/*.db | head -1)" gpurun_out/${tag}_decode_kernel_stats.txt > /dev/null
rm -rf gpurun_out/prof_$tag
cat gpurun_out/${tag}_decode_bench.log | tail -4
head -16 gpurun_out/${tag}_decode_kernel_stats.txt | cut -c1-220
