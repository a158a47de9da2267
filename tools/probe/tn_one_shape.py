"""One weight-gradient product, a few launches (for rocprofv3 --pmc runs):   python tools/probe/tn_one_shape.py [M N K]"""
import os, sys, torch
_r = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _r); sys.path.insert(0, os.path.join(_r, "tools"))
from mdvit_amd import _lib, ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 128, 65536)
A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); out = torch.zeros((M, N), device="cuda")
for _ in range(6):
    ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=False, allow_split=True, accumulate=True, precision=1)
torch.cuda.synchronize()
