#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/prof_d
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d -o d -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eval --no-other-configs > gpurun_out/r04d_bench.json 2>/dev/null
python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_d/*.db | head -1)" gpurun_out/r04d_kernel_stats.txt > /dev/null
rm -rf gpurun_out/prof_d
grep -i "pool\|cast\|kernel  \|gemm_pp4m_f16" gpurun_out/r04d_kernel_stats.txt | cut -c1-60,112-160
