#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
HVLM_GEMV_EARLY=2 timeout 900 python -m pytest tests/test_gpu_decode_r3.py -q -x 2>&1 | tail -3
for e in 1 2 1 2; do echo "== gemv_early=$e"; HVLM_GEMV_EARLY=$e timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH=0" | head -2; done
HVLM_GEMV_EARLY=2 bash tools/decode_trace.sh r6i > /dev/null 2>&1; head -9 gpurun_out/r6i_decode_kernel_stats.txt | cut -c1-200
