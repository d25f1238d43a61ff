#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 300 python3 tools/attn_clip_err.py > gpurun_out/r05x_clip_err.txt 2>&1; cat gpurun_out/r05x_clip_err.txt
for a in 0 1; do HVLM_ATTN_CLIP_ASM=$a python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; done
