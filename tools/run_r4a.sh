#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_fulldepth_r4.py -q -m gpu -s -k "test_c or test_zz" > gpurun_out/fd_c.log 2>&1; echo "c rc=$?"
cp gpurun_out/fulldepth_report.json gpurun_out/fulldepth_report_c.json
( time python bench.py ) > gpurun_out/r04a_bench7b_default.json 2> gpurun_out/r04a_bench7b_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r04a_bench7b_default.err
python3 tools/gemm_traffic_all.py drive-fp8 gpurun_out/r04a_gemm_traffic_fp8.json > gpurun_out/r04a_traffic_fp8.log 2>&1; echo "fp8 traffic rc=$?"; tail -3 gpurun_out/r04a_traffic_fp8.log
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fulldepth_r4.py 2>&1 | tail -8
