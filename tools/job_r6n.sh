#!/bin/bash
# round 6, final job: the whole GPU suite on the final build, the default bench line, the N = 2 code path of bench.py on one device (gloo), smoke
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06p_pytest.log; cat gpurun_out/r06p_pytest.log
(time python bench.py > gpurun_out/r06p_bench_default.json 2> gpurun_out/r06p_bench_default.err) 2>&1 | grep real
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06p_bench_default.json'))
print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['roofline'].get('mfma_busy_source'), d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample','samples_timed')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'], e.get('batched_fp8',{}).get('samples_per_s'), e.get('cpu_baseline',{}).get('value'))
print({k:(v.get('value'),v.get('ms_per_step'),v.get('steps')) for k,v in d['other_configs'].items()})
print(d['cpu_baseline']['value'], d['h2d']['tiling_detected'], d['h2d']['h2d_ms_exposed'])
PY
HVLM_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --layers 4 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r06p_n2_shared.err | cut -c1-600
tail -3 gpurun_out/r06p_n2_shared.err | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
