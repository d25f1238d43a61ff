#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 900 python3 tools/gemm_knob_energy.py > gpurun_out/r06e_gemm_knob_energy.txt 2>&1; cat gpurun_out/r06e_gemm_knob_energy.txt
