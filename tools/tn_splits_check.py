"""TN (weight-gradient) GEMM: time the tall-skinny shapes of the C = 64 / 128 stages over tile configs and K-splits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops

lib = _lib.load()


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in ((64, 64, 262144), (192, 64, 262144), (64, 512, 262144), (512, 64, 262144), (128, 128, 65536), (384, 128, 65536),
                  (1024, 128, 65536), (128, 1024, 65536), (320, 320, 16384), (960, 320, 16384), (1280, 320, 16384)):
    A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); out = torch.zeros((M, N), device="cuda")

    def run():
        ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=False, allow_split=True, accumulate=True, precision=1)
    lib.mdvit_gemm_force_plan(-1, 0)
    t_pl = timed(run)
    res = []
    for cfg in (0, 1, 2):
        for sp in (16, 32, 64, 128, 256, 512, 1024):
            if sp > K // 256:
                continue
            lib.mdvit_gemm_force_plan(cfg, sp)
            res.append((timed(run, 5), cfg, sp))
    lib.mdvit_gemm_force_plan(-1, 0)
    res.sort()
    hbm = 4.0 * K * (M + N) / 6e12 * 1e6
    print(f"M={M:5d} N={N:5d} K={K:7d}: planner {t_pl:7.1f} us | best " + "  ".join(f"cfg{c} sp={s}: {t:.1f}" for t, c, s in res[:4]) + f" | HBM bound {hbm:.1f} us", flush=True)
