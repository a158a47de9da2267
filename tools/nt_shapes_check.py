"""Isolated time of the step's output-heavy short-K NT GEMMs against their HBM bound (GPU-only timing behind a spin kernel)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops
lib = _lib.load()


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(6_000_000)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, full) in ((65536, 512, 64, 0), (65536, 512, 64, 1), (65536, 1024, 128, 0), (262144, 64, 64, 0), (262144, 64, 512, 0), (262144, 192, 64, 0),
                        (65536, 128, 1024, 0), (65536, 64, 512, 0), (16384, 1280, 320, 0), (16384, 320, 1280, 0), (4096, 2048, 512, 0), (4096, 512, 2048, 0)):
    x = torch.randn((M, K), device="cuda"); W = torch.randn((N, K), device="cuda"); b = torch.randn((N,), device="cuda")
    y = torch.empty((M, N), device="cuda"); res = torch.randn((M, N), device="cuda") if full else None
    res_t = {}
    for cfg in (-1, 0, 1, 2):
        lib.mdvit_gemm_force_plan(cfg, 0)
        def run():
            ops.gemm(ops._p(x), ops._p(W), ops._p(y), M, N, K, lda=K, ldb=K, ldc=N, bias=ops._p(b), e_drop=0.1 if full else 0.0, e_key=(1, 2),
                     residual=ops._p(res) if full else None, ldr=N, allow_split=True, precision=1)
        res_t[cfg] = timed(run)
    lib.mdvit_gemm_force_plan(-1, 0)
    by = 4.0 * (M * K + N * K + M * N * (2 if full else 1))
    print(f"M={M:6d} N={N:5d} K={K:5d} {'FULL ' if full else 'plain'}: planner {res_t[-1]:7.1f} us | 128x128 {res_t[0]:7.1f}  256x64 {res_t[1]:7.1f}  64x64 {res_t[2]:7.1f} | "
          f"{by / 1e6:7.1f} MB -> {by / res_t[-1] / 1e6:5.2f} TB/s, {2e-6 * M * N * K / res_t[-1]:6.1f} TF/s; HBM bound at 6 TB/s {by / 6e6:6.1f} us", flush=True)
