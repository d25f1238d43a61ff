#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2"
for r in 1 2; do
  HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_base.so $B > gpurun_out/r04f_base$r.json 2>/dev/null
  $B > gpurun_out/r04f_new$r.json 2>/dev/null
done
python - <<'PY'
import json
for f in ("base1","new1","base2","new2"):
    d=json.load(open(f"gpurun_out/r04f_{f}.json")); print(f, round(d["ms_per_step"],1), d["step_ms"], round(d["loss"],4), round(d["roofline"]["achieved"],1))
PY
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
