#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
bash tools/mfma_pmc.sh > gpurun_out/r06f_mfma_pmc_summary.txt 2>&1; cp gpurun_out/mfma_pmc.json gpurun_out/r06f_mfma_pmc.json 2>/dev/null; cat gpurun_out/r06f_mfma_pmc_summary.txt
rm -rf gpurun_out/mfma_pmc/p1 gpurun_out/mfma_pmc/t
