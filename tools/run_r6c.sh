#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HVLM_LIB=$PWD/handsonvlm-release_amd/lib/libhvlm_hip_dev.so
timeout 600 python3 tools/attn_variant_energy.py > gpurun_out/r06c_attn_variant_energy.txt 2>&1; cat gpurun_out/r06c_attn_variant_energy.txt
timeout 300 python3 tools/attn_variant_energy_clip_dma.py > gpurun_out/r06c_clip_dma_energy.txt 2>&1; cat gpurun_out/r06c_clip_dma_energy.txt
