#!/bin/bash
# round 6, job j: after gemv_early became the default — decode / r6 tests, the bench line, the decode trace
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_r6.py tests/test_gpu_decode_r3.py "tests/test_gpu_model.py" -q -x 2>&1 | tail -3
bash tools/collect_round.sh r06n bench
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06n_bench7b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'], e.get('batched_fp8',{}).get('samples_per_s'), e.get('batched',{}).get('samples_per_s'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['other_configs'].items()})
c=d['other_configs']['config2_7b_inference_336px_b1']; print(c.get('roofline',{}).get('decode'), (c.get('cpu_baseline') or {}).get('value'))
PY
bash tools/decode_trace.sh r06n > /dev/null 2>&1; head -9 gpurun_out/r06n_decode_kernel_stats.txt | cut -c1-200
