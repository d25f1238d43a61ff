#!/bin/bash
# config 4 (13B, S = 2048, B = 4) with the hand-scheduled attention forward on / off, same box, A B A B (the kernel saves joules at S = 2048)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="python bench.py --model 13b --batch 4 --text-len 1693 --no-cpu-baseline --no-eval --no-other-configs --no-h2d --steps 4 --warmup 2"
for rep in 1 2; do for asm in 0 1; do
  HVLM_ATTN_FWD_ASM=$asm $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); e=d.get('energy') or {}; print('13B S=2048 B=4  attn_fwd_asm=$asm rep=$rep  ms_per_step %.1f  J/step %.0f  step_ms %s' % (d['ms_per_step'], e.get('joules_per_step', 0), d.get('step_ms')))"
done; done > gpurun_out/r05r_13b_attn_asm_ab.txt; cat gpurun_out/r05r_13b_attn_asm_ab.txt
