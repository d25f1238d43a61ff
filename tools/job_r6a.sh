#!/bin/bash
# round 6, first GPU job: new tests, decode trace, vendor energy cross-check, baseline bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_r6.py tests/test_gpu_r5.py -q -x 2>&1 | tail -5 > gpurun_out/r6a_pytest.log; cat gpurun_out/r6a_pytest.log
bash tools/decode_trace.sh r6a
timeout 300 python3 tools/vendor_energy.py > gpurun_out/r6a_vendor_energy.txt 2> gpurun_out/r6a_vendor_energy.err; cat gpurun_out/r6a_vendor_energy.txt; tail -2 gpurun_out/r6a_vendor_energy.err
timeout 600 python bench.py > gpurun_out/r6a_bench7b.json 2> gpurun_out/r6a_bench7b.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r6a_bench7b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'])
print(json.dumps(e.get('cpu_baseline'))[:600])
print(json.dumps(d.get('cpu_baseline'))[:300])
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['other_configs'].items()})
PY
