#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
( time python bench.py --steps 10 --warmup 3 ) > gpurun_out/r04i_bench7b_default.json 2> gpurun_out/r04i_bench7b_default.err; echo "bench rc=$?"; tail -4 gpurun_out/r04i_bench7b_default.err
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -16
