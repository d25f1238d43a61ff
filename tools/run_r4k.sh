#!/bin/bash
# same-box A/B of two libraries on the step (HVLM_LIB=lib/libhvlm_base.so vs the in-tree library)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
BASE=handsonvlm-release_amd/lib/libhvlm_base.so
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2"
for r in 1 2 3; do
  HVLM_LIB=$BASE $B > gpurun_out/r04s_base$r.json 2>/dev/null
  $B > gpurun_out/r04s_new$r.json 2>/dev/null
done
python - <<'PY'
import json
for r in (1,2,3):
  for f in ("base","new"):
    d=json.load(open(f"gpurun_out/r04s_{f}{r}.json")); print(f, r, round(d["ms_per_step"],1), d["step_ms"], round(d["loss"],4), round(d["roofline"]["achieved"],1), round(d["box_calibration"]["gemm_13872x12288x4096_bf16_tflops"],1))
PY
