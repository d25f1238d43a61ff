#!/bin/bash
# round 6, job e: the persistent decode kernel (parity, timing), column tail split test, vendor kernel names
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_r6.py -q -x -k "engine or column_tail" 2>&1 | tail -12
for e in 0 1 0 1; do echo "== decode_engine=$e"; HVLM_DECODE_ENGINE=$e timeout 120 python3 tools/decode_bench.py 2>&1 | grep "GRAPH" | head -2; done
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT}"
rm -rf gpurun_out/prof_vend; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_vend -o vend -- python3 tools/vendor_energy.py --once > /dev/null 2>&1
python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_vend/*.db | head -1)" gpurun_out/r6e_vendor_kernels.txt > /dev/null; rm -rf gpurun_out/prof_vend
grep -i "cijk\|gemm_pp" gpurun_out/r6e_vendor_kernels.txt | cut -c1-260
