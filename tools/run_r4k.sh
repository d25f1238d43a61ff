#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --steps 6 --warmup 3 --dedup --no-cpu-baseline --no-eval --no-other-configs"
$B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dedup poolfc2', d['ms_per_step'], d['step_ms'], d['device_allocs_in_timed_region'])"
$B --no-pool-fc2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dedup no-poolfc2', d['ms_per_step'], d['step_ms'], d['device_allocs_in_timed_region'])"
