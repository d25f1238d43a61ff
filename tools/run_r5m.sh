#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 300 python3 tools/attn_clip_check.py > gpurun_out/r05m_clip_check.log 2>&1; tail -14 gpurun_out/r05m_clip_check.log
timeout 300 python3 tools/attn_clip_prof.py > gpurun_out/r05m_clip_prof.log 2>&1; cat gpurun_out/r05m_clip_prof.log
python -m pytest tests/test_gpu_r5.py -q -k clip 2>&1 | tail -3
# does the per-GEMM event hook of bench.py cost step time?  same box, A B A B
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --no-h2d --steps 6 --warmup 2"
for rep in 1 2; do for ev in "" "--no-gemm-events"; do
  $B $ev 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('events=%-18s rep=$rep  ms_per_step %.1f  step_ms %s' % ('$ev' or 'on', d['ms_per_step'], d.get('step_ms')))"
done; done > gpurun_out/r05m_event_hook_ab.txt; cat gpurun_out/r05m_event_hook_ab.txt
timeout 600 python3 tools/energy_probe.py > gpurun_out/r05m_energy_probe.txt 2>gpurun_out/r05m_energy_probe.err; cat gpurun_out/r05m_energy_probe.txt; tail -3 gpurun_out/r05m_energy_probe.err
