"""micro-benchmark of mdvit_gemm_f32 on the model's shapes vs torch.matmul (rocBLAS/hipBLASLt fp32) as a yardstick"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdvit_amd import ops, _lib
dev = torch.device("cuda:0")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

shapes = [(65536, 512, 64), (65536, 64, 512), (65536, 192, 64), (65536, 64, 64), (16384, 1024, 128), (16384, 128, 1024), (16384, 384, 128),
          (4096, 1280, 320), (4096, 320, 1280), (1024, 2048, 512), (1024, 512, 2048), (1024, 1536, 512)]
print(f"{'M':>6} {'N':>5} {'K':>5} | {'NT plain':>9} {'TF':>6} {'GB/s':>6} | {'+gelu2':>8} | {'+res,drop':>9} | {'NN dgrad':>9} | {'TN wgrad':>9} | {'torch NT':>9} {'torch TN':>9}")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
    g = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev)
    p = ops._p
    t_plain = timeit(lambda: ops.gemm(p(x), p(W), p(y), M, N, K, lda=K, ldb=K, ldc=N, bias=p(b)))
    t_gelu = timeit(lambda: ops.gemm(p(x), p(W), p(y), M, N, K, lda=K, ldb=K, ldc=N, bias=p(b), out2=p(y2), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2)))
    t_res = timeit(lambda: ops.gemm(p(x), p(W), p(y), M, N, K, lda=K, ldb=K, ldc=N, bias=p(b), e_drop=0.1, e_key=(1, 2), residual=p(res), ldr=N))
    t_nn = timeit(lambda: ops.gemm(p(g), p(W), p(dx), M, K, N, lda=N, ldb=K, ldc=K, trans_b=False, allow_split=True))
    t_tn = timeit(lambda: ops.gemm(p(g), p(x), p(dW), N, K, M, lda=N, ldb=K, ldc=K, trans_a=True, trans_b=False, allow_split=True))
    t_t = timeit(lambda: torch.addmm(b, x, W.t(), out=y))
    t_tt = timeit(lambda: torch.mm(g.t(), x, out=dW))
    fl = 2.0 * M * N * K; by = 4.0 * (M * K + N * K + M * N)
    print(f"{M:6d} {N:5d} {K:5d} | {t_plain:9.1f} {fl/t_plain/1e6:6.1f} {by/t_plain/1e3:6.0f} | {t_gelu:8.1f} | {t_res:9.1f} | {t_nn:9.1f} | {t_tn:9.1f} | {t_t:9.1f} {t_tt:9.1f}")
