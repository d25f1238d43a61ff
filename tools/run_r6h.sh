#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests/test_gpu_fulldepth_r5.py -q -k "test_g" 2>&1 | tail -5
python3 -c "import json; r=json.load(open('gpurun_out/fulldepth_r5_report.json'))['g_13b_L40_S2048_fwd_bwd']; print(r['inference_forward'], r['loss_rel'], r['hip_vs_fp32_oracle']['hidden'], r['bf16_eager_oracle_vs_fp32_oracle']['hidden'])"
python -m pytest tests/test_gpu_fulldepth_r4.py -q 2>&1 | tail -2
