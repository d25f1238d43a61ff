#!/bin/bash
# Reproduces round 6's evidence under gpurun_out/ on the GPU box, one section per GPU job of the round (profiles/README.md names the section
# behind every r06g - r06p file):   /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash tools/round6_jobs.sh <section>'
# Sections whose experiment was removed again (b: GEMV weights-first, d: weight-touch prefetch, i: one row per wave) ran on the commits named in
# profiles/README.md; on the final tree they re-run the surviving parts (removed switches are ignored by the library).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
case "${1:-}" in
a)  # round 6, first GPU job: new tests, decode trace, vendor energy cross-check, baseline bench
timeout 900 python -m pytest tests/test_gpu_r6.py tests/test_gpu_r5.py -q -x 2>&1 | tail -5 > gpurun_out/r6a_pytest.log; cat gpurun_out/r6a_pytest.log
bash tools/decode_trace.sh r6a
timeout 300 python3 tools/vendor_energy.py > gpurun_out/r6a_vendor_energy.txt 2> gpurun_out/r6a_vendor_energy.err; cat gpurun_out/r6a_vendor_energy.txt; tail -2 gpurun_out/r6a_vendor_energy.err
timeout 600 python bench.py > gpurun_out/r6a_bench7b.json 2> gpurun_out/r6a_bench7b.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r6a_bench7b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'])
print(json.dumps(e.get('cpu_baseline'))[:600])
print(json.dumps(d.get('cpu_baseline'))[:300])
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['other_configs'].items()})
PY
;;
b)  # round 6, job b: decode kernels (tests, trace, A/B), eval phases
timeout 1500 python -m pytest tests/test_gpu_r6.py tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" -q -x 2>&1 | tail -8 > gpurun_out/r6b_pytest.log; cat gpurun_out/r6b_pytest.log
bash tools/decode_trace.sh r6b
HVLM_DECODE_ATTN_V1=1 timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v1: /'
timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v2: /'
timeout 300 python3 tools/eval_phases.py 2>&1 | tail -1 > gpurun_out/r6b_eval_phases.json; cat gpurun_out/r6b_eval_phases.json
;;
c)  # round 6, job c: decode kernels again (tests, trace, A/B), operand-cost ablation of the GEMM (DEV library)
timeout 1500 python -m pytest tests/test_gpu_r6.py tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" -q -x 2>&1 | tail -8 > gpurun_out/r6c_pytest.log; cat gpurun_out/r6c_pytest.log
bash tools/decode_trace.sh r6c
HVLM_DECODE_ATTN_V1=1 timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v1: /'
timeout 120 python3 tools/decode_bench.py 2>&1 | grep ms/token | sed 's/^/attn_v2: /'
HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_dev.so timeout 400 python3 tools/gemm_operand_cost.py > gpurun_out/r6c_gemm_operand_cost.txt 2> gpurun_out/r6c_gemm_operand_cost.err; cat gpurun_out/r6c_gemm_operand_cost.txt; tail -3 gpurun_out/r6c_gemm_operand_cost.err
;;
d)  # round 6, job d: decode experiments: gate|up pair fusion (now default), attention split cap, weight prefetch beside the attention
timeout 900 python -m pytest tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" tests/test_gpu_r6.py -q -x -k "decode" 2>&1 | tail -4
run() { echo "== $*"; env "$@" HVLM_DECODE_GRAPH_SKIP=1 timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH=0" | head -2; }
run A=1
run HVLM_DECODE_ATTN_NS=8
run HVLM_DECODE_ATTN_NS=16
run HVLM_DECODE_PREFETCH=32
run HVLM_DECODE_PREFETCH=64
run HVLM_DECODE_PREFETCH=128
run HVLM_DECODE_PREFETCH=64 HVLM_DECODE_ATTN_NS=8
run A=1
bash tools/decode_trace.sh r6d > /dev/null 2>&1; head -9 gpurun_out/r6d_decode_kernel_stats.txt | cut -c1-200
HVLM_DECODE_PREFETCH=64 bash tools/decode_trace.sh r6d_pf > /dev/null 2>&1; head -10 gpurun_out/r6d_pf_decode_kernel_stats.txt | cut -c1-200
;;
e)  # round 6, job e: the persistent decode kernel (parity, timing), column tail split test, vendor kernel names
timeout 600 python -m pytest tests/test_gpu_r6.py -q -x -k "engine or column_tail" 2>&1 | tail -12
for e in 0 1 0 1; do echo "== decode_engine=$e"; HVLM_DECODE_ENGINE=$e timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH" | head -2; done
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT}"
rm -rf gpurun_out/prof_vend; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_vend -o vend -- python3 tools/vendor_energy.py --once > /dev/null 2>&1
python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_vend/*.db | head -1)" gpurun_out/r6e_vendor_kernels.txt > /dev/null; rm -rf gpurun_out/prof_vend
grep -i "cijk\|gemm_pp" gpurun_out/r6e_vendor_kernels.txt | cut -c1-260
;;
f)  # round 6, job f: fp8 traffic pricing, decode after the prologue change, eval phases, then the full GPU suite
timeout 300 python3 tools/fp8_traffic_price.py > gpurun_out/r6f_fp8_traffic_price.txt 2>/dev/null; cat gpurun_out/r6f_fp8_traffic_price.txt
bash tools/fp8_traffic.sh > gpurun_out/r6f_fp8_traffic_pmc.txt 2>&1; grep "==\|mean" gpurun_out/r6f_fp8_traffic_pmc.txt | cut -c1-200
timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH" | head -2
timeout 300 python3 tools/eval_phases.py 2>&1 | tail -1 > gpurun_out/r6f_eval_phases.json; cat gpurun_out/r6f_eval_phases.json
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r6f_pytest.log; cat gpurun_out/r6f_pytest.log
;;
g)  # round 6, job g: prefill column-tail A/B, then the round's evidence (bench line, kernel stats, energy probe, decode trace)
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
;;
h)
HVLM_GEMV_EARLY=1 timeout 900 python -m pytest tests/test_gpu_decode_r3.py "tests/test_gpu_model.py::test_graph_replayed_decode_bit_identical_to_eager" -q -x 2>&1 | tail -3
for e in 0 1 0 1; do echo "== gemv_early=$e"; HVLM_GEMV_EARLY=$e timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH=0" | head -2; done
HVLM_GEMV_EARLY=1 bash tools/decode_trace.sh r6h > /dev/null 2>&1; head -9 gpurun_out/r6h_decode_kernel_stats.txt | cut -c1-200
;;
i)
HVLM_GEMV_EARLY=2 timeout 900 python -m pytest tests/test_gpu_decode_r3.py -q -x 2>&1 | tail -3
for e in 1 2 1 2; do echo "== gemv_early=$e"; HVLM_GEMV_EARLY=$e timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH=0" | head -2; done
HVLM_GEMV_EARLY=2 bash tools/decode_trace.sh r6i > /dev/null 2>&1; head -9 gpurun_out/r6i_decode_kernel_stats.txt | cut -c1-200
;;
j)  # round 6, job j: after gemv_early became the default — decode / r6 tests, the bench line, the decode trace
timeout 1200 python -m pytest tests/test_gpu_r6.py tests/test_gpu_decode_r3.py "tests/test_gpu_model.py" -q -x 2>&1 | tail -3
bash tools/collect_round.sh r06n bench
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06n_bench7b.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'], e.get('batched_fp8',{}).get('samples_per_s'), e.get('batched',{}).get('samples_per_s'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['other_configs'].items()})
c=d['other_configs']['config2_7b_inference_336px_b1']; print(c.get('roofline',{}).get('decode'), (c.get('cpu_baseline') or {}).get('value'))
PY
bash tools/decode_trace.sh r06n > /dev/null 2>&1; head -9 gpurun_out/r06n_decode_kernel_stats.txt | cut -c1-200
;;
n)  # round 6, final job: the whole GPU suite on the final build, the default bench line, the N = 2 code path of bench.py on one device (gloo), smoke
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06p_pytest.log; cat gpurun_out/r06p_pytest.log
(time python bench.py > gpurun_out/r06p_bench_default.json 2> gpurun_out/r06p_bench_default.err) 2>&1 | grep real
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06p_bench_default.json'))
print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['roofline'].get('mfma_busy_source'), d.get('energy',{}).get('joules_per_step'), d['box_calibration']['gemm_13872x12288x4096_bf16_tflops'])
e=d['eval']; print({k:e[k] for k in ('samples_per_s','ms_per_sample','samples_timed')}, e['roofline']['prefill']['frac'], e['roofline']['decode']['frac'], e['roofline']['decode']['ms_per_token'], e.get('batched_fp8',{}).get('samples_per_s'), e.get('cpu_baseline',{}).get('value'))
print({k:(v.get('value'),v.get('ms_per_step'),v.get('steps')) for k,v in d['other_configs'].items()})
print(d['cpu_baseline']['value'], d['h2d']['tiling_detected'], d['h2d']['h2d_ms_exposed'])
PY
HVLM_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --layers 4 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r06p_n2_shared.err | cut -c1-600
tail -3 gpurun_out/r06p_n2_shared.err | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
;;
*) echo "usage: bash tools/round6_jobs.sh <a|b|c|d|e|f|g|h|i|j|n>"; exit 2 ;;
esac
