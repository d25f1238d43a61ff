#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python3 - <<'PY'
import sys, torch
sys.path.insert(0, "handsonvlm-release_amd")
from hvlm import ops, _lib
for (B,S,H) in [(2,867,4),(1,2048,3),(3,300,2)]:
    qkv = (torch.randn(B*S, 3*H*128, device="cuda")*0.5).to(torch.bfloat16)
    _lib.configure("attn_fwd_pipe", 0); o0,l0 = ops.attn_fwd(qkv,B,S,H,128,True,128**-0.5)
    _lib.configure("attn_fwd_pipe", 1); o1,l1 = ops.attn_fwd(qkv,B,S,H,128,True,128**-0.5)
    print(B,S,H,"bit-identical:", torch.equal(o0,o1), torch.equal(l0,l1), float((o0.float()-o1.float()).abs().max()))
_lib.configure("attn_fwd_pipe", 0)
PY
echo "== baseline"; python3 tools/attn_bench.py --all 2>&1 | grep "causal=1"
echo "== pipelined (1 workgroup per CU, 356 VGPRs)"; HVLM_ATTN_FWD_PIPE=1 python3 tools/attn_bench.py --all 2>&1 | grep "causal=1"
