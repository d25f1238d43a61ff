#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 600 python3 tools/gemm_shape_probe.py > gpurun_out/r05t_gemm_shape_probe.txt 2>&1; cat gpurun_out/r05t_gemm_shape_probe.txt
