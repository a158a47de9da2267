"""One plane-GEMM shape, a few launches (the workload of tools/pmc_stalls.sh):  python tools/probe/gemm_ph_one_shape.py M N K [cfg] [a_f32] [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd._lib import call  # noqa: E402
from gemm_bp_check import planes_of, run_bp  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 1280, 1280)
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
a_f32 = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
out = torch.empty((M, N), device="cuda")
xp, wp = planes_of(x), planes_of(w)
call("mdvit_gemm_planes_force_plan", cfg, 0)
for _ in range(iters):
    run_bp(x if a_f32 else xp, wp, M, N, K, a_f32=a_f32, C_out=out)
torch.cuda.synchronize()
