"""A handful of GEMM launches for a rocprofv3 --pmc pass (stall breakdown of the short-K kernels)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import _lib, ops
lib = _lib.load()
M = 262144
def run(N, K, cfg, ta=False, epi=0):
    if ta:
        A = torch.randn((K, N), device="cuda"); B = torch.randn((K, 64), device="cuda"); out = torch.empty((N, 64), device="cuda")
        lib.mdvit_gemm_force_plan(cfg, 128); ops._plan_cache.clear()
        for _ in range(3):
            ops.gemm(ops._p(A), ops._p(B), ops._p(out), N, 64, K, lda=N, ldb=64, ldc=64, trans_a=True, trans_b=False, allow_split=True)
        return
    A = torch.randn((M, K), device="cuda"); B = torch.randn((N, K), device="cuda"); out = torch.empty((M, N), device="cuda")
    lib.mdvit_gemm_force_plan(cfg, 1); ops._plan_cache.clear()
    for _ in range(3):
        ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, trans_a=False, trans_b=True)
def run_rc(Hd, Cn):
    """the MLP's fc2 data-gradient with the recomputed pre-activation (EPI_DGELU_RC)"""
    g = torch.randn((M, Cn), device="cuda"); W2t = torch.randn((Hd, Cn), device="cuda"); du = torch.empty((M, Hd), device="cuda")
    x = torch.randn((M, Cn), device="cuda"); W1 = torch.randn((Hd, Cn), device="cuda"); b1 = torch.randn(Hd, device="cuda")
    lib.mdvit_gemm_force_plan(-1, 0); ops._plan_cache.clear()
    for _ in range(3):
        ops.gemm(ops._p(g), ops._p(W2t), ops._p(du), M, Hd, Cn, lda=Cn, ldb=Cn, ldc=Hd, trans_a=False, trans_b=True, epi=_lib.EPI_DGELU,
                 rc=(ops._p(x), Cn, ops._p(W1), Cn, ops._p(b1), Cn), e_drop=0.1, e_key=(1, 2), precision=1)
run_rc(512, 64)
run(512, 64, 2); run(512, 64, 0); run(64, 512, 2); run(64, 512, 1)
run(512, 262144, 2, ta=True); run(512, 262144, 0, ta=True)
torch.cuda.synchronize()
