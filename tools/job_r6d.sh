#!/bin/bash
# round 6, job d: decode experiments: gate|up pair fusion (now default), attention split cap, weight prefetch beside the attention
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" tests/test_gpu_r6.py -q -x -k "decode" 2>&1 | tail -4
run() { echo "== $*"; env "$@" HVLM_DECODE_GRAPH_SKIP=1 timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH=0" | head -2; }
run A=1
run HVLM_DECODE_ATTN_NS=8
run HVLM_DECODE_ATTN_NS=16
run HVLM_DECODE_PREFETCH=32
run HVLM_DECODE_PREFETCH=64
run HVLM_DECODE_PREFETCH=128
run HVLM_DECODE_PREFETCH=64 HVLM_DECODE_ATTN_NS=8
run A=1
bash tools/decode_trace.sh r6d > /dev/null 2>&1; head -9 gpurun_out/r6d_decode_kernel_stats.txt | cut -c1-200
HVLM_DECODE_PREFETCH=64 bash tools/decode_trace.sh r6d_pf > /dev/null 2>&1; head -10 gpurun_out/r6d_pf_decode_kernel_stats.txt | cut -c1-200
