#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
HVLM_GEMV_EARLY=1 timeout 900 python -m pytest tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" -q -x 2>&1 | tail -3
for e in 0 1 0 1; do echo "== gemv_early=$e"; HVLM_GEMV_EARLY=$e timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH=0" | head -2; done
HVLM_GEMV_EARLY=1 bash tools/decode_trace.sh r6h > /dev/null 2>&1; head -9 gpurun_out/r6h_decode_kernel_stats.txt | cut -c1-200
