#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests -m gpu -q --durations=14 2>&1 | tail -24 > gpurun_out/r06g_pytest_durations.log; cat gpurun_out/r06g_pytest_durations.log
