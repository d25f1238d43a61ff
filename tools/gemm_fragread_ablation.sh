#!/bin/bash
# Round-3 verdict item 4 (first experiment), as an ABLATION with counters instead of a new kernel: what would a 1-wave-per-SIMD
# 128x128 wave tile buy?  Its whole benefit over the shipped 128x64 wave tile is fewer LDS->VGPR fragment reads per MFMA
# ((128+128)/(128*128) vs (128+64)/(128*64): one third fewer).  The development build's K-loop ablation switch gemm_dbg = 8 removes
# exactly a third of the fragment reads of the shipped schedule (stale A fragments of row half 1: WRONG results, timing / counters
# only) and changes nothing else — an upper bound for what the bigger wave tile could gain, without its costs (one wave per SIMD:
# no partner wave to cover its LDS waits; 256 accumulators in AccVGPRs).  Shapes: the decoder's q|k|v and down_proj GEMMs.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gemm_fragread_ablation.sh'
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HVLM_LIB=handsonvlm-release_amd/lib/libhvlm_hip_dev.so
OUT=gpurun_out/r04h_gemm_fragread_ablation.txt; mkdir -p gpurun_out; : > $OUT
for shape in "13872 12288 4096" "13872 4096 11008"; do
  for dbg in 0 8; do
    echo "== $shape  gemm_m16=0 gemm_dbg=$dbg" | tee -a $OUT
    HVLM_GEMM_M16=0 HVLM_GEMM_DBG=$dbg python3 - $shape <<'PY' | tee -a $OUT
import os, sys, torch
sys.path.insert(0, "handsonvlm-release_amd")
from hvlm import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
a = torch.rand(M, K, device="cuda").sub_(0.5).to(torch.bfloat16); b = torch.rand(N, K, device="cuda").sub_(0.5).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(5): ops.gemm_nt(a, b, out=out)
torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    s.record()
    for _ in range(20): ops.gemm_nt(a, b, out=out)
    e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) / 20)
print(f"time {best*1e3:.1f} us  {2.0*M*N*K/best/1e9:.0f} TFLOP/s")
PY
    for ctr in SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY; do
      d=gpurun_out/pmc_fr_$ctr; rm -rf $d
      HVLM_GEMM_M16=0 HVLM_GEMM_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -- python3 tools/gemm_one.py $shape 3 > /dev/null 2>&1
      python3 - $d $ctr <<'PY' | tee -a $OUT
import csv, glob, sys
d, ctr = sys.argv[1:3]
v = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "gemm_pp" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
print(f"  {ctr:32s} per launch: {sum(v)/max(1,len(v)):.4e}  (n={len(v)})")
PY
      rm -rf $d
    done
  done
done
