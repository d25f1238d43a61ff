#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
BASE=handsonvlm-release_amd/lib/libhvlm_base.so
for r in 1 2; do
  echo "== base"; HVLM_LIB=$BASE python3 tools/clip_epi_bench.py 2>&1 | grep -v amdgpu.ids
  echo "== new";  python3 tools/clip_epi_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04q_clip_epi_ab.txt
HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_prof.so python3 tools/gemm_tile_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04p_gemm_tile_timeline.txt
grep -E "^==|first_ktile|tile period|pre_epi|post_epi|epilogue  " gpurun_out/r04p_gemm_tile_timeline.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_r2.py tests/test_gpu_r4.py -q -x -m gpu 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2"
for r in 1 2; do
  HVLM_LIB=$BASE $B > gpurun_out/r04q_base$r.json 2>/dev/null
  $B > gpurun_out/r04q_new$r.json 2>/dev/null
done
python - <<'PY'
import json
for f in ("base1","new1","base2","new2"):
    d=json.load(open(f"gpurun_out/r04q_{f}.json")); print(f, round(d["ms_per_step"],1), d["step_ms"], round(d["loss"],4), round(d["roofline"]["achieved"],1))
PY
