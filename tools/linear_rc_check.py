"""mdvit_linear_rc (streaming short-K Linear) against mdvit_gemm_f32 on the step's shapes: results and timing.   python tools/linear_rc_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import ops
from mdvit_amd.ops import call, _p, _stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
shapes = [(262144, 192, 64, False), (262144, 64, 64, True), (262144, 64, 64, False), (524288, 192, 64, False), (65536, 384, 128, False), (65536, 128, 128, True),
          (131072, 384, 128, False), (65536, 512, 64, True), (65536, 512, 64, False), (16384, 512, 128, True), (4173, 96, 64, True)]
for M, N, K, full in shapes:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, N, device="cuda") if full else None
    rs = ((torch.rand(4, device="cuda") < 0.9).float() / 0.9) if full else None
    rps = (M + 3) // 4
    p = 0.1 if full else 0.0
    y0, y1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    Wp = torch.empty(2, N, K, device="cuda", dtype=torch.bfloat16)
    call("mdvit_split_planes_t", _p(W), K, _p(Wp), K, N * K, N, K, 0, 2, _stream())

    def ref():
        ops.gemm(_p(x), _p(W), _p(y0), M, N, K, lda=K, ldb=K, ldc=N, bias=_p(b), e_drop=p, e_key=(11, 22), e_rowscale=_p(rs), e_rows_per_scale=rps,
                 residual=_p(res), ldr=N)

    def new():
        call("mdvit_linear_rc", _p(x), K, _p(Wp), N * K, _p(b), _p(y1), N, M, N, K, p, 11, 22, _p(rs), rps, _p(res), N, None, _stream())
    ref(); new()
    err = float((y0 - y1).abs().max() / y0.abs().max())
    t0, t1 = timed(ref), timed(new)
    byts = 4.0 * (M * K + M * N * (2 if full else 1))
    print(f"M={M:7d} N={N:4d} K={K:4d} full={int(full)}: gemm {t0:7.1f} us ({byts / t0 / 1e6:5.2f} TB/s)   linear_rc {t1:7.1f} us ({byts / t1 / 1e6:5.2f} TB/s)   max rel diff {err:.2e}", flush=True)
