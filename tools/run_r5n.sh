#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2 > gpurun_out/r05n_bench_energy.json 2> gpurun_out/r05n_bench_energy.err
python3 -c "import json; d=json.load(open('gpurun_out/r05n_bench_energy.json')); print(d['ms_per_step'], d.get('energy'))"
python -m pytest tests/test_gpu_r5.py -q 2>&1 | tail -2
bash tools/traffic_r5.sh > gpurun_out/r05n_traffic.txt 2>&1; cat gpurun_out/r05n_traffic.txt
rm -rf gpurun_out/traffic_r5/*/*/*.db; find gpurun_out/traffic_r5 -name "*.csv" -size +2M -delete
