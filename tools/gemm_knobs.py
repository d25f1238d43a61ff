"""GEMM knob sweep (GPU box): environment switches that need no rebuild, on the SFT step's main shapes."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "handsonvlm-release_amd"))
from hvlm import ops
BF = torch.bfloat16
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
out = {}
for (M, N, K) in [(13872, 12288, 4096), (13872, 4096, 4096), (13872, 22016, 4096), (102800, 3072, 1024), (102800, 1024, 4096), (12288, 4096, 13888)]:
    a = torch.rand(M, K, device="cuda").sub_(0.5).to(BF); b = torch.rand(N, K, device="cuda").sub_(0.5).to(BF)
    c = torch.empty(M, N, dtype=BF, device="cuda")
    best = min(timeit(lambda: ops.gemm_nt(a, b, out=c)) for _ in range(2))
    out["%%dx%%dx%%d" %% (M, N, K)] = round(2.0 * M * N * K / best / 1e9)
    del a, b, c
print(out)
''' % ROOT
ENVS = [{}, {"HVLM_GEMM_M16": "0"}, {}, {"HVLM_GEMM_M16": "0"}] if len(sys.argv) < 2 else [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]]
for env in ENVS:
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True)
    print(env, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
