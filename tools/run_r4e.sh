#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_r2.py tests/test_gpu_fp8.py -q -m gpu 2>&1 | tail -8
echo "== base"; HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_base.so python3 tools/clip_epi_bench.py
echo "== new";  python3 tools/clip_epi_bench.py
echo "== base"; HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_base.so python3 tools/clip_epi_bench.py
echo "== new";  python3 tools/clip_epi_bench.py
