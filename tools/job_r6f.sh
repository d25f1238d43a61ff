#!/bin/bash
# round 6, job f: fp8 traffic pricing, decode after the prologue change, eval phases, then the full GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 300 python3 tools/fp8_traffic_price.py > gpurun_out/r6f_fp8_traffic_price.txt 2>/dev/null; cat gpurun_out/r6f_fp8_traffic_price.txt
bash tools/fp8_traffic.sh > gpurun_out/r6f_fp8_traffic_pmc.txt 2>&1; grep "==\|mean" gpurun_out/r6f_fp8_traffic_pmc.txt | cut -c1-200
timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH" | head -2
timeout 300 python3 tools/eval_phases.py 2>&1 | tail -1 > gpurun_out/r6f_eval_phases.json; cat gpurun_out/r6f_eval_phases.json
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r6f_pytest.log; cat gpurun_out/r6f_pytest.log
