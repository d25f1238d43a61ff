#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 600 tools/build/mfma_power_probe 0 > gpurun_out/r05s_mfma_power_probe_duty.txt 2>&1; cat gpurun_out/r05s_mfma_power_probe_duty.txt
