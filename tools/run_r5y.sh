#!/bin/bash
# which of the two assembly kernels carries the step's 1.4 ms?  one switch at a time, same box, A B A B A B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --no-h2d --no-gemm-events --steps 8 --warmup 2"
for rep in 1 2 3; do for cfg in "1 1" "1 0" "0 1" "0 0"; do
  set -- $cfg
  HVLM_ATTN_FWD_ASM=$1 HVLM_ATTN_CLIP_ASM=$2 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); e=d.get('energy') or {}; print('fwd_asm=$1 clip_asm=$2 rep=$rep  ms_per_step %.1f  J/step %.0f' % (d['ms_per_step'], e.get('joules_per_step', 0)))"
done; done > gpurun_out/r05y_asm_switches_ab.txt; cat gpurun_out/r05y_asm_switches_ab.txt
