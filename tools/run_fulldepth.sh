#!/bin/bash
# Full-depth parity tests (tests/test_gpu_fulldepth_r4.py) alone, with durations; report -> gpurun_out/fulldepth_report.json
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fulldepth_r4.py -q -m gpu -s --durations=10 > gpurun_out/fd_full.log 2>&1; echo "full rc=$?"; tail -25 gpurun_out/fd_full.log
