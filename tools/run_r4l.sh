#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_r2.py tests/test_gpu_model.py -q -m gpu -k "attn or attention or layer or forward_backward or packed or recompute" 2>&1 | tail -4
echo "== delta inside dQ"; python3 tools/attn_bench.py --all 2>&1 | grep "causal=1"
echo "== separate delta pass"; HVLM_ATTN_DELTA_SEPARATE=1 python3 tools/attn_bench.py --all 2>&1 | grep "causal=1"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2"
for r in 1 2; do
  HVLM_ATTN_DELTA_SEPARATE=1 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('separate', round(d['ms_per_step'],1), d['step_ms'])"
  $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused   ', round(d['ms_per_step'],1), d['step_ms'])"
done
