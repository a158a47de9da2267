"""PROBE: gemm_pm's time with k-tile-major weight planes ([K/32][N][32]: every 1 KiB LDS-DMA piece one contiguous KiB) against row-major planes (64-byte pieces of K-long rows).
Timing only -- the products are garbage in the tiled run (the kernel takes ldb == 32 as the layout flag of this probe build)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops, _lib
from mdvit_amd._lib import call, PlaneGemmDesc
from gemm_bp_check import planes_of

def run(x, wp, out, M, N, K, ldb):
    d = PlaneGemmDesc()
    d.A = ops._p(x); d.lda = K; d.a_plane = 0; d.a_f32 = 1
    d.B = ops._p(wp); d.ldb = ldb; d.b_plane = N * K
    d.planes = 2; d.trans = 0; d.M, d.N, d.K = M, N, K
    d.C = ops._p(out); d.ldc = N
    call("mdvit_gemm_planes", C.byref(d), ops._stream())

def timed(fn, n=20):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (M, N, K) in ((16384, 320, 1280), (16384, 320, 960), (16384, 320, 320), (32768, 320, 1280), (131072, 320, 1280), (16384, 1280, 320), (8192, 512, 2048)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
    out = torch.empty(M, N, device="cuda"); wp = planes_of(w)
    call("mdvit_gemm_planes_force_plan", 6 if N % 160 == 0 else 7, 0)
    t_row = timed(lambda: run(x, wp, out, M, N, K, K))
    t_tiled = timed(lambda: run(x, wp, out, M, N, K, 32))
    print(f"{M:7d} x {N:5d} x {K:5d}: row-major planes {t_row:7.1f} us   k-tile-major planes {t_tiled:7.1f} us", flush=True)
