cd "${GRAFT_REPO_ROOT:-/root/repo}"
for g in 4 2 8 16 1; do echo "== gemm_group_m=$g"; HVLM_GEMM_GROUP_M=$g python3 tools/clip_epi_bench.py 2>/dev/null | grep -v "^$"; done
python3 tools/gemm_knobs.py "HVLM_GEMM_GROUP_M=4" "HVLM_GEMM_GROUP_M=8" "HVLM_GEMM_GROUP_M=2" "HVLM_GEMM_GROUP_M=16" 2>&1 | tail -5
