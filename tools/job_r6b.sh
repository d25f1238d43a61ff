#!/bin/bash
# round 6, job b: decode kernels (tests, trace, A/B), eval phases
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_r6.py tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" -q -x 2>&1 | tail -8 > gpurun_out/r6b_pytest.log; cat gpurun_out/r6b_pytest.log
bash tools/decode_trace.sh r6b
HVLM_DECODE_ATTN_V1=1 timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v1: /'
timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v2: /'
timeout 300 python3 tools/eval_phases.py 2>&1 | tail -1 > gpurun_out/r6b_eval_phases.json; cat gpurun_out/r6b_eval_phases.json
