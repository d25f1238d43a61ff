#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 300 python3 tools/attn_clip_check.py > gpurun_out/r06d_clip_check.txt 2>&1; tail -10 gpurun_out/r06d_clip_check.txt
python -m pytest tests/test_gpu_r5.py tests/test_gpu_parity_r2.py tests/test_gpu_r4.py tests/test_gpu_model.py tests/test_gpu_fulldepth_r4.py -q 2>&1 | tail -3 > gpurun_out/r06d_pytest_subset.log; cat gpurun_out/r06d_pytest_subset.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
