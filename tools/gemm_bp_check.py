"""Plane GEMM (mdvit_gemm_planes) on the GPU: correctness against fp64 and timing against mdvit_gemm_f32 (bf16x3) on the
step's key NT shapes.   python tools/gemm_bp_check.py [--quick]"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops  # noqa: E402
from mdvit_amd._lib import PlaneGemmDesc, call  # noqa: E402


def planes_of(x, planes=2):
    """fp32 [R, K] -> bf16 planes tensor [planes, R, K] via the library's splitter"""
    R, K = x.shape
    out = torch.empty((planes, R, K), device=x.device, dtype=torch.bfloat16)
    call("mdvit_split_planes", ops._p(x), K, ops._p(out), K, R * K, R, K, planes, ops._stream())
    return out


def join(pl):
    return pl.float().sum(0)


def run_bp(A, Bp, M, N, K, *, a_f32, planes=2, bias=None, epi=0, C_out=None, Cp=None, U=None, residual=None, rowscale=None, rps=1,
           drop=0.0, key=(0, 0), gelu_u=None, rc=None, allow_split=False, accumulate=False):
    d = PlaneGemmDesc()
    d.A = ops._p(A); d.lda = K; d.a_plane = 0 if a_f32 else M * K; d.a_f32 = int(a_f32)
    d.B = ops._p(Bp); d.ldb = K; d.b_plane = N * K
    d.planes = planes; d.trans = 0; d.M, d.N, d.K = M, N, K
    if C_out is not None:
        d.C = ops._p(C_out); d.ldc = N
    if Cp is not None:
        d.Cp = ops._p(Cp); d.ldcp = N; d.c_plane = M * N
    if U is not None:
        d.U = ops._p(U); d.ldu_out = N
    d.bias = ops._p(bias); d.epi = epi
    d.e_drop_p = drop; d.e_key0, d.e_key1 = key
    d.e_rowscale = ops._p(rowscale); d.e_rows_per_scale = rps
    d.residual = ops._p(residual); d.ldr = N
    d.gelu_u = ops._p(gelu_u); d.ldu = N
    if rc is not None:
        ra, rb, rbias, rk = rc
        d.rc_a = ops._p(ra); d.rc_lda = rk; d.rc_a_plane = M * rk
        d.rc_b = ops._p(rb); d.rc_ldb = rk; d.rc_b_plane = N * rk
        d.rc_bias = ops._p(rbias); d.rc_k = rk
    d.allow_split = int(allow_split); d.accumulate = int(accumulate)
    ws = None
    if allow_split:
        need = _lib.load().mdvit_gemm_planes_ws_bytes(C.byref(d))
        if need:
            ws = torch.empty((need // 4,), device="cuda", dtype=torch.float32)
            d.ws, d.ws_bytes = ops._p(ws), need
    call("mdvit_gemm_planes", C.byref(d), ops._stream())
    return ws


def gelu(x):
    return torch.nn.functional.gelu(x)


def gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


def check(tag, got, want, tol):
    err = float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
    ok = err <= tol
    print(f"  {'ok ' if ok else 'BAD'} {tag}: rel-to-max err {err:.2e} (tol {tol:.0e})", flush=True)
    return ok


def correctness():
    torch.manual_seed(0)
    ok = True
    for cfg in (-1, 0, 1, 2):
        call("mdvit_gemm_planes_force_plan", cfg, 0)
        for (M, N, K) in ((300, 200, 64), (1000, 72, 320), (130, 64, 1024), (64, 512, 128)):
            x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
            b = torch.randn(N, device="cuda")
            ref = x.double() @ w.double().T + b.double()
            xp, wp = planes_of(x), planes_of(w)
            for a_f32 in (False, True):
                out = torch.full((M, N), float("nan"), device="cuda")
                outp = torch.zeros((2, M, N), device="cuda", dtype=torch.bfloat16)
                run_bp(x if a_f32 else xp, wp, M, N, K, a_f32=a_f32, bias=b, C_out=out, Cp=outp)
                ok &= check(f"cfg {cfg} plain {M}x{N}x{K} a_f32={int(a_f32)}", out, ref, 2e-5)
                ok &= check(f"cfg {cfg}   planes out", join(outp), ref, 3e-5)
            # bf16 speed mode
            out = torch.empty((M, N), device="cuda")
            run_bp(planes_of(x, 1), planes_of(w, 1), M, N, K, a_f32=False, planes=1, bias=b, C_out=out)
            ok &= check(f"cfg {cfg} bf16 single plane {M}x{N}x{K}", out, ref, 2e-2)
            # GELU with U
            U = torch.empty((M, N), device="cuda"); h = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, bias=b, epi=_lib.EPI_GELU_DUAL, C_out=h, U=U)
            ok &= check(f"cfg {cfg} gelu u", U, ref, 2e-5)
            ok &= check(f"cfg {cfg} gelu h", h, gelu(ref), 2e-5)
            # FULL: rowscale + residual (no dropout)
            res = torch.randn(M, N, device="cuda"); rs = torch.rand((M + 49) // 50, device="cuda")
            out = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=out, residual=res, rowscale=rs, rps=50)
            want = res.double() + rs.double().repeat_interleave(50)[:M, None] * ref
            ok &= check(f"cfg {cfg} full", out, want, 2e-5)
            # DGELU with u from HBM
            u = torch.randn(M, N, device="cuda")
            out = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, epi=_lib.EPI_DGELU, C_out=out, gelu_u=u)
            ok &= check(f"cfg {cfg} dgelu", out, (x.double() @ w.double().T) * gelu_grad(u.double()), 2e-5)
            # split-K + accumulate
            if K >= 512:
                call("mdvit_gemm_planes_force_plan", cfg, 2)
                acc0 = torch.randn(M, N, device="cuda"); out = acc0.clone()
                run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=out, allow_split=True, accumulate=True)
                ok &= check(f"cfg {cfg} split-K + accumulate", out, ref + acc0.double(), 2e-5)
                call("mdvit_gemm_planes_force_plan", cfg, 0)
        if cfg in (1, 2):
            # DGELU with the pre-activation recomputed
            M, N, K, RK = 500, 256, 64, 64
            g = torch.randn(M, K, device="cuda"); w2t = torch.randn(N, K, device="cuda") * 0.1
            x = torch.randn(M, RK, device="cuda"); w1 = torch.randn(N, RK, device="cuda") * 0.2; b1 = torch.randn(N, device="cuda")
            out = torch.empty((M, N), device="cuda")
            run_bp(planes_of(g), planes_of(w2t), M, N, K, a_f32=False, epi=_lib.EPI_DGELU, C_out=out, rc=(planes_of(x), planes_of(w1), b1, RK))
            u = x.double() @ w1.double().T + b1.double()
            ok &= check(f"cfg {cfg} dgelu recompute", out, (g.double() @ w2t.double().T) * gelu_grad(u), 2e-5)
    call("mdvit_gemm_planes_force_plan", -1, 0)
    return ok


def time_it(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [  # (M, N, K, note)   bs=4 fused (16 images) and bs=32 block shapes
    (262144, 192, 64, "qkv s0 bs4"), (262144, 64, 64, "proj s0 bs4"), (262144, 64, 192, "qkv dgrad s0"), (262144, 512, 64, "fc1 s0"),
    (262144, 64, 512, "fc2 s0"), (65536, 384, 128, "qkv s1"), (65536, 128, 128, "proj s1"), (65536, 1024, 128, "fc1 s1"), (65536, 128, 1024, "fc2 s1"),
    (16384, 960, 320, "qkv s2"), (16384, 320, 320, "proj s2"), (16384, 1280, 320, "fc1 s2"), (16384, 320, 1280, "fc2 s2"),
    (4096, 1536, 512, "qkv s3"), (4096, 512, 512, "proj s3"), (4096, 2048, 512, "fc1 s3"), (4096, 512, 2048, "fc2 s3"),
    (4096, 1024, 4608, "bridge"), (4096, 4608, 1024, "bridge dgrad"),
    (32768, 960, 320, "qkv s2 bs32"), (32768, 1280, 320, "fc1 s2 bs32"), (32768, 320, 1280, "fc2 s2 bs32"), (8192, 2048, 512, "fc1 s3 bs32"),
    (524288, 192, 64, "qkv s0 bs32"), (131072, 1024, 128, "fc1 s1 bs32"),
]


def timing(quick):
    print("\nshape                          old bf16x3 |  planes a_f32 | planes (glds)  [cfg sweep: best cfg] | bf16 1-plane | HBM-bound us")
    for (M, N, K, note) in SHAPES[: 8 if quick else None]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
        out = torch.empty((M, N), device="cuda")
        xp, wp = planes_of(x), planes_of(w)
        t_old = time_it(lambda: ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1))
        res = {}
        for cfg in (0, 1, 2):
            call("mdvit_gemm_planes_force_plan", cfg, 0)
            res[("f32", cfg)] = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
            res[("pl", cfg)] = time_it(lambda: run_bp(xp, wp, M, N, K, a_f32=False, C_out=out))
        call("mdvit_gemm_planes_force_plan", -1, 0)
        t_auto = time_it(lambda: run_bp(xp, wp, M, N, K, a_f32=False, C_out=out))
        xp1, wp1 = planes_of(x, 1), planes_of(w, 1)
        t_1 = time_it(lambda: run_bp(xp1, wp1, M, N, K, a_f32=False, planes=1, C_out=out))
        bf = min((res[("f32", c)], c) for c in (0, 1, 2)); bp = min((res[("pl", c)], c) for c in (0, 1, 2))
        hbm = 4.0 * (M * K + N * K + M * N) / 6.0e12 * 1e6
        fl = 2.0 * M * N * K
        print(f"{note:14s} {M:7d}x{N:5d}x{K:5d}  {t_old:7.1f}us {fl / t_old / 1e6:6.1f}TF | {bf[0]:7.1f}us c{bf[1]} | {bp[0]:7.1f}us c{bp[1]} {fl / bp[0] / 1e6:6.1f}TF "
              f"(auto {t_auto:7.1f}) [{res[('pl', 0)]:.0f} {res[('pl', 1)]:.0f} {res[('pl', 2)]:.0f}] | {t_1:7.1f}us | {hbm:6.1f}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    a = ap.parse_args()
    good = correctness()
    print("CORRECTNESS", "PASS" if good else "FAIL", flush=True)
    if not a.no_timing:
        timing(a.quick)
    sys.exit(0 if good else 1)
