#!/bin/bash
# Reproduces the round's evidence under gpurun_out/ on the GPU box (copy what you want judged into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/collect_round.sh r02x'
# Sections can be selected: bash tools/collect_round.sh r02x tests bench variants stats pmc probe
set -u
tag=${1:-rXX}; shift || true
what=${*:-tests bench variants stats}
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --steps 6 --warmup 2"
for w in $what; do case $w in
  tests)    python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${tag}_pytest.log ;;
  bench)    python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench7b.json 2> gpurun_out/${tag}_bench7b.err ;;
  variants) $B --dedup > gpurun_out/${tag}_bench7b_dedup.json 2>/dev/null
            $B --varlen 64:192 > gpurun_out/${tag}_varlen_padded.json 2>/dev/null
            $B --varlen 64:192 --pack > gpurun_out/${tag}_varlen_packed.json 2>/dev/null
            $B --recompute > gpurun_out/${tag}_7b_recompute.json 2>/dev/null
            $B --image-size 336 > gpurun_out/${tag}_bench7b_336px.json 2>/dev/null
            $B --model 13b --batch 4 --text-len 1693 --steps 3 --warmup 1 > gpurun_out/${tag}_13b_b4.json 2>/dev/null
            $B --model 13b --batch 8 --text-len 1693 --recompute 26 --steps 3 --warmup 1 > gpurun_out/${tag}_13b_b8_rc26.json 2>/dev/null
            $B --model 13b --batch 16 --text-len 1693 --recompute 36 --steps 3 --warmup 1 > gpurun_out/${tag}_13b_b16_rc36.json 2>/dev/null
            python bench.py --fp8-eval --steps 2 --warmup 1 > gpurun_out/${tag}_fp8_eval.json 2>/dev/null ;;
  stats)    rm -rf gpurun_out/prof_$tag
            rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-other-configs > /dev/null 2>&1
            python3 tools/rocpd_stats.py "$(ls gpurun_out/prof_$tag/*.db | head -1)" gpurun_out/${tag}_bench7b_kernel_stats.txt > /dev/null
            rm -rf gpurun_out/prof_$tag ;;
  pmc)      python bench.py --steps 2 --warmup 1 --gemm-table --no-cpu-baseline --no-eval --no-other-configs > /dev/null 2>&1     # -> gpurun_out/gemm_by_shape.json
            python3 tools/gemm_traffic_all.py drive gpurun_out/gemm_by_shape.json gpurun_out/${tag}_gemm_traffic.json > gpurun_out/${tag}_traffic.log 2>&1
            # (two rocprofv3 --pmc passes over all shapes; copy the .json to profiles/: bench.py picks the newest profiles/r*_gemm_traffic.json that covers its shapes)
            python3 tools/gemm_traffic_all.py drive-fp8 gpurun_out/${tag}_gemm_traffic_fp8.json > gpurun_out/${tag}_traffic_fp8.log 2>&1
            ;;
  probe)    tools/build/mfma_power_probe 0 > gpurun_out/${tag}_mfma_power_probe.txt; tools/build/mfma_power_probe 1 >> gpurun_out/${tag}_mfma_power_probe.txt ;;
esac; done
ls -la gpurun_out | grep "$tag" | awk '{print $5, $9}'
