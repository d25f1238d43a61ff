#!/bin/bash
# Builds lib/libhvlm_hip_prof.so = the production objects + gemm_pp.hip compiled with -DHVLM_GEMM_PROF (s_memtime stamps of the tile
# cycle, see HVLM_PP_STAMP).  Run tools/gemm_tile_timeline.py with HVLM_LIB pointing at it (on the GPU box).
set -e
cd "$(dirname "$0")/../handsonvlm-release_amd"
make -j8 >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DHVLM_GEMM_PROF -c csrc/gemm_pp.hip -o build/gemm_pp_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhvlm_hip_prof.so $(ls build/*.o | grep -v -e gemm_pp.o -e gemm_pp_prof.o) build/gemm_pp_prof.o
echo built lib/libhvlm_hip_prof.so
