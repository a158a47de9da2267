"""Why are the output-heavy short-K GEMMs at ~50 % of the pure-store rate?  single vs dual output, power-of-two vs padded
distance between the two output arrays.   PYTHONPATH=. python tools/probe/gemm_store.py"""
import torch
from mdvit_amd import _lib, ops

lib = _lib.load()


def timed(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, N, K = 262144, 512, 64
A = torch.randn((M, K), device="cuda"); B = torch.randn((N, K), device="cuda"); bias = torch.randn(N, device="cuda")
pool = torch.empty((2 * M * N + (1 << 22),), device="cuda")
out = pool[:M * N].view(M, N)
for name, off in (("out2 adjacent (2^29 B apart)", M * N), ("out2 +4352 B pad", M * N + 1088), ("out2 +1 MiB + 768 B pad", M * N + (1 << 18) + 192)):
    out2 = pool[off:off + M * N].view(M, N)
    for c, cn in ((0, "128x128"), (2, "64x64")):
        lib.mdvit_gemm_force_plan(c, 1); ops._plan_cache.clear()
        t = timed(lambda: ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, trans_a=False, trans_b=True,
                                   out2=ops._p(out2), bias=ops._p(bias), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2)))
        print(f"gelu dual  {name:30s} {cn:8s} {t:7.1f} us  {4.0 * (M * K + 2 * M * N) / t / 1e6:5.2f} TB/s")
for c, cn in ((0, "128x128"), (1, "256x64"), (2, "64x64")):
    lib.mdvit_gemm_force_plan(c, 1); ops._plan_cache.clear()
    t = timed(lambda: ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, trans_a=False, trans_b=True, bias=ops._p(bias)))
    print(f"plain single output                       {cn:8s} {t:7.1f} us  {4.0 * (M * K + M * N) / t / 1e6:5.2f} TB/s")
    t = timed(lambda: ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, trans_a=False, trans_b=True, bias=ops._p(bias), precision=0))
    print(f"plain single output, fp32 MFMA            {cn:8s} {t:7.1f} us  {4.0 * (M * K + M * N) / t / 1e6:5.2f} TB/s")
t = timed(lambda: out.fill_(1.0))
print(f"torch fill of the output                           {t:7.1f} us  {4.0 * M * N / t / 1e6:5.2f} TB/s")
t = timed(lambda: torch.mul(out, 2.0, out=pool[M * N:2 * M * N].view(M, N)))
print(f"torch out2 = 2*out (read + write)                  {t:7.1f} us  {8.0 * M * N / t / 1e6:5.2f} TB/s")
