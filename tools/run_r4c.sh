#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2"
$B > gpurun_out/r04c_poolfc2_on.json 2>/dev/null
$B --no-pool-fc2 > gpurun_out/r04c_poolfc2_off.json 2>/dev/null
$B > gpurun_out/r04c_poolfc2_on2.json 2>/dev/null
$B --no-pool-fc2 > gpurun_out/r04c_poolfc2_off2.json 2>/dev/null
python - <<'PY'
import json
for f in ("on","off","on2","off2"):
    d=json.load(open(f"gpurun_out/r04c_poolfc2_{f}.json")); print(f, round(d["ms_per_step"],1), d["step_ms"], d["device_allocs_in_timed_region"], round(d["loss"],4), round(d["mfma_roofline_frac_whole_step"],4))
PY
