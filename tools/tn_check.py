"""Transposing-read wgrad kernel (gemm_tn.hip): correctness vs fp64 and timing against the general template it replaces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops

lib = _lib.load()


def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(6_000_000)        # ~3 ms of GPU idle spin: the host enqueues all n launches behind it, so e0..e1 is GPU time only
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(A, B, out, M, N, K, lda=None, ldb=None, accumulate=False, colsum=None, precision=1):
    ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=lda or M, ldb=ldb or N, ldc=N, trans_a=True, trans_b=False, allow_split=True,
             accumulate=accumulate, precision=precision, colsum_a=ops._p(colsum) if colsum is not None else None)


torch.manual_seed(0)
bad = 0
for (M, N, K) in ((64, 64, 1000), (320, 192, 4100), (128, 128, 777), (1280, 320, 2048), (192, 64, 33), (64, 512, 5000), (512, 64, 8192), (100, 36, 515)):
    for cfg in (-1, 0, 1, 2, 3):
        for sp in (0, 1, 3):
            lib.mdvit_gemm_tn_config(1, cfg, sp)
            A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda")
            out0 = torch.randn((M, N), device="cuda"); out = out0.clone()
            cs0 = torch.randn((M,), device="cuda"); cs = cs0.clone()
            run(A, B, out, M, N, K, accumulate=True, colsum=cs)
            ref = out0.double() + A.double().t() @ B.double()
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            cerr = float((cs.double() - (cs0.double() + A.double().sum(0))).abs().max() / max(1.0, float(A.double().sum(0).abs().max())))
            ok = err < 2e-5 and cerr < 1e-5
            bad += not ok
            if not ok or (cfg == -1 and sp == 0):
                print(f"M={M} N={N} K={K} cfg={cfg} sp={sp}: err {err:.2e} colsum err {cerr:.2e} {'ok' if ok else 'BAD'}", flush=True)
# strided operands (lda > M: a column block of a wider tensor), no accumulate, bf16 single plane
lib.mdvit_gemm_tn_config(1, -1, 0)
K, M, N = 3000, 128, 64
Aw = torch.randn((K, 384), device="cuda"); Bw = torch.randn((K, 256), device="cuda")
out = torch.full((M, N), float("nan"), device="cuda")
run(Aw[:, 128:256], Bw[:, 64:128], out, M, N, K, lda=384, ldb=256)
ref = Aw[:, 128:256].double().t() @ Bw[:, 64:128].double()
err = float((out.double() - ref).abs().max() / ref.abs().max()); print(f"strided: err {err:.2e}"); bad += not err < 2e-5
out = torch.full((M, N), float("nan"), device="cuda")
run(Aw[:, 128:256], Bw[:, 64:128], out, M, N, K, lda=384, ldb=256, precision=2)
err = float((out.double() - ref).abs().max() / ref.abs().max()); print(f"strided, one bf16 plane: err {err:.2e}"); bad += not err < 2e-2
print("correctness:", "ALL OK" if bad == 0 else f"{bad} BAD")

if "--time" in sys.argv:
    shapes = ((64, 64, 262144), (192, 64, 262144), (64, 512, 262144), (512, 64, 262144), (128, 128, 65536), (384, 128, 65536),
              (1024, 128, 65536), (128, 1024, 65536), (320, 320, 16384), (960, 320, 16384), (1280, 320, 16384), (320, 1280, 16384),
              (512, 512, 4096), (1536, 512, 4096), (2048, 512, 4096), (512, 2048, 4096))
    for (M, N, K) in shapes:
        A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); out = torch.zeros((M, N), device="cuda")
        lib.mdvit_gemm_tn_config(0, -1, 0)
        t_old = timed(lambda: run(A, B, out, M, N, K, accumulate=True))
        lib.mdvit_gemm_tn_config(1, -1, 0)
        t_new = timed(lambda: run(A, B, out, M, N, K, accumulate=True))
        t_one = timed(lambda: run(A, B, out, M, N, K, accumulate=True, precision=2))
        res = []
        if "--sweep" in sys.argv:
            for cfg in (0, 1, 2, 3):
                for sp in (4, 8, 16, 32, 64, 128, 256, 512, 1024):
                    if sp > K // 256:
                        continue
                    lib.mdvit_gemm_tn_config(1, cfg, sp)
                    res.append((timed(lambda: run(A, B, out, M, N, K, accumulate=True), 4), cfg, sp))
            res.sort()
        lib.mdvit_gemm_tn_config(1, -1, 0)
        hbm = 4.0 * K * (M + N) / 8e12 * 1e6
        mf = 2.0 * M * N * K / (2500e12 / 3) * 1e6
        print(f"M={M:5d} N={N:5d} K={K:7d}: old {t_old:7.1f} us  new {t_new:7.1f} us ({2e-6 * M * N * K / t_new:6.1f} TF/s)  one-plane {t_one:7.1f} | "
              + "  ".join(f"cfg{c} sp={s}: {t:.1f}" for t, c, s in res[:4]) + f" | bound hbm {hbm:.1f} mfma {mf:.1f} us", flush=True)
