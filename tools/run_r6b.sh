#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 300 python3 tools/attn_asm_check.py --time > gpurun_out/r06b_attn_fwd_check.txt 2>&1; tail -8 gpurun_out/r06b_attn_fwd_check.txt
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r06b_pytest.log; cat gpurun_out/r06b_pytest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r06b_bench7b.json 2> gpurun_out/r06b_bench7b.err
python3 -c "import json; d=json.load(open('gpurun_out/r06b_bench7b.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])"
