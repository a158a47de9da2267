"""Ring GEMMs on the GPU: correctness vs fp64 and timing vs the other kernels.  python tools/ring_check.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops
from mdvit_amd._lib import call
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bp_check import planes_of, run_bp, check, time_it, gelu, gelu_grad

lib = _lib.load()


def correctness():
    ok = True
    torch.manual_seed(1)
    for ring_cfg in (0, 1):
        call("mdvit_gemm_ring_config", 1, ring_cfg)
        for (M, N, K) in ((1000, 256, 64), (520, 384, 320), (300, 128, 1024), (2048, 1280, 96)):
            x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
            ref = x.double() @ w.double().T + b.double()
            wp = planes_of(w)
            out = torch.full((M, N), float("nan"), device="cuda"); outp = torch.zeros((2, M, N), device="cuda", dtype=torch.bfloat16)
            run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=out, Cp=outp)
            ok &= check(f"ring nt cfg {ring_cfg} plain {M}x{N}x{K}", out, ref, 2e-5)
            ok &= check("   planes out", outp.float().sum(0), ref, 3e-5)
            U = torch.empty((M, N), device="cuda"); h = torch.empty((M, N), device="cuda")
            run_bp(x, wp, M, N, K, a_f32=True, bias=b, epi=_lib.EPI_GELU_DUAL, C_out=h, U=U)
            ok &= check("   gelu u", U, ref, 2e-5); ok &= check("   gelu h", h, gelu(ref), 2e-5)
            res = torch.randn(M, N, device="cuda"); rs = torch.rand((M + 49) // 50, device="cuda")
            run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=out, residual=res, rowscale=rs, rps=50)
            ok &= check("   full", out, res.double() + rs.double().repeat_interleave(50)[:M, None] * ref, 2e-5)
            u = torch.randn(M, N, device="cuda")
            run_bp(x, wp, M, N, K, a_f32=True, epi=_lib.EPI_DGELU, C_out=out, gelu_u=u)
            ok &= check("   dgelu", out, (x.double() @ w.double().T) * gelu_grad(u.double()), 2e-5)
            out1 = torch.empty((M, N), device="cuda")
            run_bp(x, planes_of(w, 1), M, N, K, a_f32=True, planes=1, bias=b, C_out=out1)
            ok &= check("   bf16 single plane", out1, ref, 2e-2)
            if K >= 1024:
                acc0 = torch.randn(M, N, device="cuda"); o2 = acc0.clone()
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=o2, allow_split=True, accumulate=True)
                ok &= check("   split-K + accumulate", o2, ref + acc0.double(), 2e-5)
    call("mdvit_gemm_ring_config", 1, -1)
    # TN
    for (M, N, K) in ((128, 128, 4096), (384, 128, 8192), (320, 320, 2048), (1280, 320, 1024), (132, 260, 512)):
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda")
        ref = A.double().T @ B.double()
        for split in (False, True):
            acc0 = torch.randn(M, N, device="cuda"); out = acc0.clone(); cs = torch.zeros(M, device="cuda")
            ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=False, allow_split=split, accumulate=True,
                     precision=1, colsum_a=ops._p(cs))
            ok &= check(f"ring tn {M}x{N}x{K} split={split}", out, ref + acc0.double(), 2e-5)
            ok &= check("   colsum", cs, A.double().sum(0), 2e-5)
    return ok


NT = [(16384, 960, 320, "qkv s2"), (16384, 1280, 320, "fc1 s2"), (16384, 320, 1280, "fc2 s2"), (4096, 1536, 512, "qkv s3"), (4096, 2048, 512, "fc1 s3"),
      (4096, 512, 2048, "fc2 s3"), (4096, 1024, 4608, "bridge"), (4096, 4608, 1024, "bridge dg"), (65536, 384, 128, "qkv s1"), (65536, 1024, 128, "fc1 s1"),
      (65536, 128, 1024, "fc2 s1"), (262144, 512, 64, "aux q0"), (262144, 192, 64, "qkv s0"),
      (32768, 960, 320, "qkv s2 bs32"), (32768, 1280, 320, "fc1 s2 bs32"), (32768, 320, 1280, "fc2 s2 bs32"), (8192, 2048, 512, "fc1 s3 bs32"),
      (131072, 1024, 128, "fc1 s1 bs32"), (131072, 384, 128, "qkv s1 bs32")]
TN = [(128, 128, 65536), (384, 128, 65536), (1024, 128, 65536), (128, 1024, 65536), (320, 320, 16384), (960, 320, 16384), (1280, 320, 16384), (320, 1280, 16384),
      (512, 512, 4096), (1536, 512, 4096), (2048, 512, 4096), (512, 2048, 4096), (1024, 4608, 4096),
      (1280, 320, 32768), (320, 1280, 32768), (2048, 512, 8192), (1024, 128, 131072)]


def timing():
    print("\nNT (fp32 A, plane B): old bf16x3 | plane v1 (a_f32) | ring 256x128 | ring 128x128")
    for (M, N, K, note) in NT:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; out = torch.empty((M, N), device="cuda")
        wp = planes_of(w)
        t_old = time_it(lambda: ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1))
        call("mdvit_gemm_ring_config", 0, -1)
        t_v1 = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
        ts = []
        for cfg in (0, 1):
            call("mdvit_gemm_ring_config", 1, cfg)
            ts.append(time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out)))
        call("mdvit_gemm_ring_config", 1, -1)
        fl = 2.0 * M * N * K
        print(f"{note:12s} {M:7d}x{N:5d}x{K:5d}  {t_old:7.1f}us {fl / t_old / 1e6:6.1f}TF | {t_v1:7.1f} | {ts[0]:7.1f}us {fl / ts[0] / 1e6:6.1f}TF | {ts[1]:7.1f}us {fl / ts[1] / 1e6:6.1f}TF", flush=True)
    print("\nTN (weight gradients): old bf16x3 (planner) | ring 128x128")
    for (M, N, K) in TN:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); out = torch.zeros((M, N), device="cuda")

        def run():
            ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=M, ldb=N, ldc=N, trans_a=True, trans_b=False, allow_split=True, accumulate=True, precision=1)
        call("mdvit_gemm_ring_config", 0, -1)
        t_old = time_it(run)
        call("mdvit_gemm_ring_config", 1, -1)
        t_new = time_it(run)
        best = (t_new, 0)
        for sp in (2, 4, 8, 16, 32, 64, 128, 256):
            if sp > K // 256:
                break
            lib.mdvit_gemm_force_plan(-1, sp)
            t = time_it(run, 8)
            if t < best[0]:
                best = (t, sp)
        lib.mdvit_gemm_force_plan(-1, 0)
        fl = 2.0 * M * N * K
        print(f"M={M:5d} N={N:5d} K={K:7d}  {t_old:7.1f}us {fl / t_old / 1e6:6.1f}TF | {t_new:7.1f}us {fl / t_new / 1e6:6.1f}TF | best forced split {best[1]}: {best[0]:.1f}us", flush=True)


if __name__ == "__main__":
    good = correctness()
    print("CORRECTNESS", "PASS" if good else "FAIL", flush=True)
    timing()
    sys.exit(0 if good else 1)
