#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05q_smoke.log 2>&1; tail -2 gpurun_out/r05q_smoke.log
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r05q_bench_driver_cmd.json 2> gpurun_out/r05q_bench_driver_cmd.err ) 2>&1 | grep real
python3 -c "import json; d=json.load(open('gpurun_out/r05q_bench_driver_cmd.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])"
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r05q
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05q -o r05q -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-other-configs --no-h2d > /dev/null 2>&1
python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_r05q/*.db | head -1)" gpurun_out/r05q_bench7b_kernel_stats.txt | head -4
rm -rf gpurun_out/prof_r05q
