"""MFMA-bound NT shapes of the late stages for a rocprofv3 --pmc pass"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import _lib, ops
lib = _lib.load()
def run(M, N, K, cfg):
    A = torch.randn((M, K), device="cuda"); B = torch.randn((N, K), device="cuda"); out = torch.empty((M, N), device="cuda")
    lib.mdvit_gemm_force_plan(cfg, 1); ops._plan_cache.clear()
    for _ in range(3):
        ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, trans_a=False, trans_b=True)
run(16384, 1280, 320, 0); run(16384, 320, 1280, 0); run(4096, 2048, 512, 0); run(4096, 512, 2048, 2); run(32768, 1280, 320, 0); run(8192, 2048, 512, 0)
torch.cuda.synchronize()
