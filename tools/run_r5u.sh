#!/bin/bash
# q|k|v GEMM -> CLIP attention through a MALL-sized buffer (HVLM_CLIP_QKV_CHUNK frames per chunk): joules and time per step, same box; the loss must not move
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --no-h2d --no-gemm-events --steps 5 --warmup 2"
for c in 0 100 64 128 50 0 100; do
  HVLM_CLIP_QKV_CHUNK=$c $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); e=d.get('energy') or {}; print('qkv_chunk=%-4s ms_per_step %.1f  J/step %.0f  W %.0f  loss %.6f' % ('$c', d['ms_per_step'], e.get('joules_per_step', 0), e.get('avg_power_w', 0), d['loss']))"
done > gpurun_out/r05u_clip_qkv_chunk.txt; cat gpurun_out/r05u_clip_qkv_chunk.txt
