#!/bin/bash
# round 6, job g: prefill column-tail A/B, then the round's evidence (bench line, kernel stats, energy probe, decode trace)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
for t in 1 0 1 0; do echo -n "HVLM_GEMM_TAIL=$t: "; HVLM_GEMM_TAIL=$t timeout 200 python3 tools/eval_phases.py 2>&1 | tail -1; done > gpurun_out/r06m_prefill_tail_ab.txt; cat gpurun_out/r06m_prefill_tail_ab.txt
bash tools/collect_round.sh r06m bench stats
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06m_bench7b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'], e.get('batched_fp8',{}).get('samples_per_s'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['other_configs'].items()})
PY
timeout 600 python3 tools/energy_probe.py > gpurun_out/r06m_energy_probe.txt 2>gpurun_out/r06m_energy_probe.err; cat gpurun_out/r06m_energy_probe.txt
bash tools/decode_trace.sh r06m > /dev/null 2>&1; head -9 gpurun_out/r06m_decode_kernel_stats.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
