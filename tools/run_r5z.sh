#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HVLM_LIB=$PWD/handsonvlm-release_amd/lib/libhvlm_hip_dev.so
timeout 300 python3 tools/attn_clip_err.py > gpurun_out/r05z_clip_err_dotsum.txt 2>&1; cat gpurun_out/r05z_clip_err_dotsum.txt
timeout 300 python3 tools/attn_clip_prof.py > gpurun_out/r05z_clip_prof_dotsum.txt 2>&1; grep -E "c\+\+|production|prof |dotsum|split" gpurun_out/r05z_clip_prof_dotsum.txt
timeout 400 python3 tools/attn_asm_prof.py > gpurun_out/r05z_fwd_prof_dotsum.txt 2>&1; grep -E "==|c\+\+|production|prof |dotsum|krows|error" gpurun_out/r05z_fwd_prof_dotsum.txt
