#!/bin/bash
# In-situ same-box A/B of the two hand-scheduled attention kernels (HVLM_ATTN_FWD_ASM / HVLM_ATTN_CLIP_ASM = 0 / 1): kernel-trace
# statistics of two bench steps per setting (per-kernel average in the step) and the step time of 6 untraced steps, A B A B.
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tag=${1:-r05i}
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --no-h2d --steps 6 --warmup 2"
for rep in 1 2; do for asm in 0 1; do
  HVLM_ATTN_FWD_ASM=$asm HVLM_ATTN_CLIP_ASM=$asm $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('asm=$asm rep=$rep  ms_per_step %.1f  step_ms %s' % (d['ms_per_step'], d.get('step_ms')))"
done; done > gpurun_out/${tag}_step_ab.txt
for asm in 0 1; do
  rm -rf gpurun_out/prof_$tag
  HVLM_ATTN_FWD_ASM=$asm HVLM_ATTN_CLIP_ASM=$asm rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-other-configs --no-h2d > /dev/null 2>&1
  python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_$tag/*.db | head -1)" gpurun_out/${tag}_kernel_stats_asm$asm.txt > /dev/null
  rm -rf gpurun_out/prof_$tag
done
cat gpurun_out/${tag}_step_ab.txt
grep -E "hvlm_fa_|attn_fwd" gpurun_out/${tag}_kernel_stats_asm*.txt | cut -c1-200
