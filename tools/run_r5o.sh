#!/bin/bash
# CLIP tower pass size vs HBM bytes: does a pass whose activations fit the 256 MB MALL (<= ~100 frames: qkv 158 MB, MLP hidden 210 MB) save the
# joules its smaller GEMMs (tile-count quantisation) cost?  Same box, bench step + energy, A B ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="python bench.py --no-cpu-baseline --no-eval --no-other-configs --no-h2d --steps 5 --warmup 2"
for c in 1600 100 64 128 200 400 1600; do
  $B --clip-chunk $c 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); e=d.get('energy') or {}; print('clip_chunk=%-5s ms_per_step %.1f  J/step %.0f  W %.0f  gemm_tflops %.0f' % ('$c', d['ms_per_step'], e.get('joules_per_step', 0), e.get('avg_power_w', 0), d['roofline']['achieved']))"
done > gpurun_out/r05o_clip_chunk.txt; cat gpurun_out/r05o_clip_chunk.txt
