"""Times the HBM-bound short-K GEMMs of the step (stage 0/1 MLP, qkv, proj; forward and data-gradient forms) per tile
config, with algorithmic TB/s.  usage: PYTHONPATH=. python tools/probe/gemm_shortk.py [rows_stage0]"""
import sys
import torch
from mdvit_amd import _lib, ops

lib = _lib.load()
CFG = {0: "128x128", 1: "256x64", 2: "64x64"}


def timed(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    shapes = []
    for T, C, Hd in ((T0, 64, 512), (T0 // 4, 128, 1024), (T0 // 16, 320, 1280), (T0 // 64, 512, 2048)):
        shapes += [(T, Hd, C, 1, "fc1+gelu"), (T, C, Hd, 3, "fc2+res"), (T, Hd, C, 2, "fc2 dgrad gelu'"), (T, C, Hd, 0, "fc1 dgrad"),
                   (T, 3 * C, C, 0, "qkv"), (T, C, 3 * C, 0, "qkv dgrad"), (T, C, C, 3, "proj+res")]
    for M, N, K, epi, name in shapes:
        A = torch.randn((M, K), device="cuda"); B = torch.randn((N, K), device="cuda"); out = torch.empty((M, N), device="cuda")
        extra, nio = {}, 1
        if epi == 1:        # single store (gelu(u) only), as the C <= 128 MLPs run it; the wide stages add out2
            bias = torch.randn(N, device="cuda")
            extra = dict(bias=ops._p(bias), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2))
            if K > 128:
                out2 = torch.empty_like(out); nio = 2
                extra["out2"] = ops._p(out2)
        elif epi == 2:
            u = torch.randn((M, N), device="cuda"); nio = 2
            extra = dict(epi=_lib.EPI_DGELU, gelu_u=ops._p(u), ldu=N, e_drop=0.1, e_key=(1, 2))
        elif epi == 3:
            res = torch.randn((M, N), device="cuda"); bias = torch.randn(N, device="cuda"); nio = 2
            extra = dict(residual=ops._p(res), ldr=N, bias=ops._p(bias), e_drop=0.1, e_key=(1, 2))
        by = 4.0 * (M * K + N * K + nio * M * N)

        def run():
            ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, trans_a=False, trans_b=True, **extra)
        res_ = []
        for c in (0, 1, 2):
            lib.mdvit_gemm_force_plan(c, 1)
            ops._plan_cache.clear()
            t = timed(run)
            res_.append(f"{CFG[c]} {t:7.1f} us {by / t / 1e6:5.2f} TB/s")
        lib.mdvit_gemm_force_plan(-1, 0)
        ops._plan_cache.clear()
        t = timed(run)
        print(f"{name:16s} M={M:7d} N={N:5d} K={K:5d} | " + " | ".join(res_) + f" | planner {t:7.1f} us {by / t / 1e6:5.2f} TB/s", flush=True)


main()
