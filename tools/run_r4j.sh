#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_next_rows.py tests/test_gpu_r4.py -q -m gpu 2>&1 | tail -4
python bench.py --steps 2 --warmup 1 --gemm-table --no-cpu-baseline --no-eval --no-other-configs > /dev/null 2>&1
python3 tools/gemm_traffic_all.py drive gpurun_out/gemm_by_shape.json gpurun_out/r04i_gemm_traffic.json > gpurun_out/r04i_traffic.log 2>&1; tail -2 gpurun_out/r04i_traffic.log
python bench.py --steps 4 --warmup 2 --dedup --no-cpu-baseline --no-eval --no-other-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dedup', d['ms_per_step'], d['config']['workload'])"
