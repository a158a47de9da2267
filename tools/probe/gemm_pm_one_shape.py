"""One NT product on the 128-row phase-split tile (csrc/gemm_pm.hip), repeated -- a target for rocprofv3 --pmc (tools/pmc_stalls.sh gemm_pm ...):
    python tools/probe/gemm_pm_one_shape.py [M N K] [iters]          default 16384 320 1280, 20 launches"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd._lib import call  # noqa: E402
from gemm_bp_check import planes_of, run_bp  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 320, 1280)
iters = int(sys.argv[4]) if len(sys.argv) >= 5 else 20
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
out = torch.empty(M, N, device="cuda")
wp = planes_of(w)
call("mdvit_gemm_planes_force_plan", 6 if N % 160 == 0 else 7, 0)
for _ in range(iters):
    run_bp(x, wp, M, N, K, a_f32=True, C_out=out)
torch.cuda.synchronize()
print("done", M, N, K, iters)
