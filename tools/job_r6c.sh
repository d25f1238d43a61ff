#!/bin/bash
# round 6, job c: decode kernels again (tests, trace, A/B), operand-cost ablation of the GEMM (DEV library)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_r6.py tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" -q -x 2>&1 | tail -8 > gpurun_out/r6c_pytest.log; cat gpurun_out/r6c_pytest.log
bash tools/decode_trace.sh r6c
HVLM_DECODE_ATTN_V1=1 timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v1: /'
timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v2: /'
HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_dev.so timeout 400 python3 tools/gemm_operand_cost.py > gpurun_out/r6c_gemm_operand_cost.txt 2> gpurun_out/r6c_gemm_operand_cost.err; cat gpurun_out/r6c_gemm_operand_cost.txt; tail -3 gpurun_out/r6c_gemm_operand_cost.err
