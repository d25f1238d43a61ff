#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests -m gpu -q -k "clip or tower or r5" 2>&1 | tail -3
( time python bench.py > gpurun_out/r05w_bench_default.json 2> gpurun_out/r05w_bench_default.err ) 2>&1 | grep real
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05w_bench_default.json'))
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'))
for k,v in d.get('other_configs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'), v.get('wall_s'))
print({k:(v.get('samples_per_s') if isinstance(v,dict) else v) for k,v in d.get('eval',{}).items()})
PY
