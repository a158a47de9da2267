"""Is the 256-wide plane GEMM's main loop held by its operand loads?  Same launch with lda = ldb = 0 (every row of a tile aliases ONE cached row: the loads
always hit, the arithmetic and the instruction stream are unchanged; the results are garbage by design).   python tools/probe/gemm_ph_cached_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops  # noqa: E402
from mdvit_amd._lib import PlaneGemmDesc, call  # noqa: E402
from gemm_bp_check import planes_of, time_it  # noqa: E402

for (M, N, K) in ((32768, 1280, 1280), (32768, 1024, 4608), (8192, 2048, 512)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
    out = torch.empty((M, N), device="cuda")
    xp, wp = planes_of(x), planes_of(w)
    call("mdvit_gemm_planes_force_plan", 3, 0)
    res = []
    for a_f32 in (True, False):
        for ld0 in (0, 1, 2, 3):           # 0: real operands, 1: A cached, 2: B cached, 3: both
            d = PlaneGemmDesc()
            d.A = ops._p(x if a_f32 else xp); d.lda = 0 if (ld0 & 1) else K; d.a_plane = 0 if a_f32 else M * K; d.a_f32 = int(a_f32)
            d.B = ops._p(wp); d.ldb = 0 if (ld0 & 2) else K; d.b_plane = N * K
            d.planes = 2; d.M, d.N, d.K = M, N, K
            d.C = ops._p(out); d.ldc = N
            t = time_it(lambda: call("mdvit_gemm_planes", C.byref(d), ops._stream()))
            res.append((a_f32, ld0, t))
    print(f"{M}x{N}x{K}: " + "  ".join(f"[a_f32={int(a)} cached={int(c)}] {t:7.1f} us ({2.0 * M * N * K / t / 1e6:4.0f} TF)" for a, c, t in res), flush=True)
