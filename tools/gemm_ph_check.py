"""Phase-split 256-wide plane GEMM (csrc/gemm_ph.hip; mdvit_gemm_planes with the plan forced to cfg >= 3) on the GPU:
correctness against fp64, BIT equality with the 128x128 plane tile, a race screen (repeat runs must agree bit for bit), and
timing against the 128-tile plane kernels and the split-while-staging kernels of gemm.hip on the step's MFMA-bound shapes.
    python tools/gemm_ph_check.py [--quick] [--no-timing] [--cfgs 3,4,5]"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdvit_amd import _lib, ops  # noqa: E402
from mdvit_amd._lib import call  # noqa: E402
from gemm_bp_check import planes_of, join, run_bp, gelu, gelu_grad, check, time_it  # noqa: E402


def force(cfg, splits=0):
    call("mdvit_gemm_planes_force_plan", cfg, splits)


def correctness(cfgs):
    torch.manual_seed(0)
    ok = True
    shapes = ((512, 512, 64), (300, 200, 128), (1000, 328, 320), (257, 260, 1024), (2048, 1280, 320), (777, 960, 320), (4096, 320, 1280))
    for cfg in cfgs:
        for (M, N, K) in shapes:
            x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
            b = torch.randn(N, device="cuda")
            # asymmetric, transpose-detecting data: a ramp on top of the noise
            x += torch.arange(M, device="cuda", dtype=torch.float32)[:, None] * 1e-3
            w += torch.arange(N, device="cuda", dtype=torch.float32)[:, None] * 1e-4
            ref = x.double() @ w.double().T + b.double()
            xp, wp = planes_of(x), planes_of(w)
            force(0)
            base = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=base)
            force(cfg)
            out = torch.full((M, N), float("nan"), device="cuda")
            outp = torch.zeros((2, M, N), device="cuda", dtype=torch.bfloat16)
            run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=out, Cp=outp)
            ok &= check(f"cfg {cfg} plain {M}x{N}x{K}", out, ref, 2e-5)
            same = bool(torch.equal(out, base))
            print(f"  {'ok ' if same else 'BAD'} cfg {cfg}   == the 128x128 plane tile bit for bit: {same}", flush=True)
            ok &= same
            ok &= check(f"cfg {cfg}   planes out", join(outp), ref, 3e-5)
            # race screen: 20 repeats, bit-identical
            rep_ok = True
            for _ in range(20):
                o2 = torch.full((M, N), float("nan"), device="cuda")
                run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=o2)
                rep_ok &= bool(torch.equal(o2, out))
            print(f"  {'ok ' if rep_ok else 'BAD'} cfg {cfg}   20 repeats identical: {rep_ok}", flush=True)
            ok &= rep_ok
            # fp32 A (split while staged)
            o3 = torch.full((M, N), float("nan"), device="cuda")
            try:
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=o3)
                same = bool(torch.equal(o3, base))
                print(f"  {'ok ' if same else 'BAD'} cfg {cfg}   fp32 A == plane A bit for bit: {same}", flush=True)
                ok &= same
            except RuntimeError as e:
                print(f"  --  cfg {cfg}   fp32 A: not built ({str(e)[:60]})", flush=True)
            # single plane
            if K % 64 == 0 and K >= 128:
                o1 = torch.empty((M, N), device="cuda"); b1 = torch.empty((M, N), device="cuda")
                xp1, wp1 = planes_of(x, 1), planes_of(w, 1)
                run_bp(xp1, wp1, M, N, K, a_f32=False, planes=1, bias=b, C_out=o1)
                force(0); run_bp(xp1, wp1, M, N, K, a_f32=False, planes=1, bias=b, C_out=b1); force(cfg)
                same = bool(torch.equal(o1, b1))
                print(f"  {'ok ' if same else 'BAD'} cfg {cfg}   single plane == 128x128 tile bit for bit: {same}", flush=True)
                ok &= same
            # epilogues
            U = torch.empty((M, N), device="cuda"); h = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, bias=b, epi=_lib.EPI_GELU_DUAL, C_out=h, U=U, drop=0.1, key=(3, 4))
            U0 = torch.empty((M, N), device="cuda"); h0 = torch.empty((M, N), device="cuda")
            force(0); run_bp(xp, wp, M, N, K, a_f32=False, bias=b, epi=_lib.EPI_GELU_DUAL, C_out=h0, U=U0, drop=0.1, key=(3, 4)); force(cfg)
            ok &= check(f"cfg {cfg} gelu u", U, ref, 2e-5)
            same = bool(torch.equal(h, h0) and torch.equal(U, U0))
            print(f"  {'ok ' if same else 'BAD'} cfg {cfg}   gelu + dropout == 128x128 tile: {same}", flush=True)
            ok &= same
            res = torch.randn(M, N, device="cuda"); rs = torch.rand((M + 49) // 50, device="cuda")
            o4 = torch.empty((M, N), device="cuda"); o5 = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=o4, residual=res, rowscale=rs, rps=50, drop=0.1, key=(5, 6))
            force(0); run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=o5, residual=res, rowscale=rs, rps=50, drop=0.1, key=(5, 6)); force(cfg)
            same = bool(torch.equal(o4, o5))
            print(f"  {'ok ' if same else 'BAD'} cfg {cfg}   full epilogue == 128x128 tile: {same}", flush=True)
            ok &= same
            u = torch.randn(M, N, device="cuda")
            o6 = torch.empty((M, N), device="cuda")
            run_bp(xp, wp, M, N, K, a_f32=False, epi=_lib.EPI_DGELU, C_out=o6, gelu_u=u)
            ok &= check(f"cfg {cfg} dgelu", o6, (x.double() @ w.double().T) * gelu_grad(u.double()), 2e-5)
            if K >= 512:
                force(cfg, 2)
                acc0 = torch.randn(M, N, device="cuda"); o7 = acc0.clone()
                run_bp(xp, wp, M, N, K, a_f32=False, bias=b, C_out=o7, allow_split=True, accumulate=True)
                ok &= check(f"cfg {cfg} split-K + accumulate", o7, ref + acc0.double(), 2e-5)
                force(cfg)
    force(-1)
    return ok


SHAPES = [
    (32768, 1280, 1280, "sweep target"),
    (32768, 960, 320, "qkv s2 bs32"), (32768, 320, 320, "proj s2 bs32"), (32768, 1280, 320, "fc1 s2 bs32"), (32768, 320, 1280, "fc2 s2 bs32"),
    (32768, 320, 960, "qkv dgrad s2"),
    (8192, 1536, 512, "qkv s3 bs32"), (8192, 512, 512, "proj s3 bs32"), (8192, 2048, 512, "fc1 s3 bs32"), (8192, 512, 2048, "fc2 s3 bs32"),
    (8192, 512, 1536, "qkv dgrad s3"),
    (16384, 1280, 320, "fc1 s2 bs4"), (16384, 320, 1280, "fc2 s2 bs4"), (4096, 2048, 512, "fc1 s3 bs4"), (4096, 512, 2048, "fc2 s3 bs4"),
    (32768, 1024, 4608, "bridge bs32"), (131072, 1280, 320, "fc1 s2 step32"),
]


def timing(quick, cfgs, variants=(1,)):
    cfgs = [(c, v) for c in cfgs for v in variants]
    print("\nshape                                   gemm.hip bf16x3 | planes 128-tiles (a_f32 / planes) | " + " | ".join(f"cfg {c} v{v} (a_f32 / planes)" for c, v in cfgs) +
          " | useful TF best", flush=True)
    for (M, N, K, note) in SHAPES[: 6 if quick else None]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
        out = torch.empty((M, N), device="cuda")
        xp, wp = planes_of(x), planes_of(w)
        fl = 2.0 * M * N * K
        force(-1)
        t_old = time_it(lambda: ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1))
        best128 = [1e9, 1e9]
        for cfg in (0, 1, 2):
            force(cfg)
            best128[0] = min(best128[0], time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out)))
            best128[1] = min(best128[1], time_it(lambda: run_bp(xp, wp, M, N, K, a_f32=False, C_out=out)))
        cols = []
        best = min(t_old, best128[0])
        for cfg, var in cfgs:
            force(cfg)
            call("mdvit_gemm_ph_config", var)
            try:
                ta = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
            except RuntimeError:
                ta = float("nan")
            try:
                tp = time_it(lambda: run_bp(xp, wp, M, N, K, a_f32=False, C_out=out))
            except RuntimeError:
                tp = float("nan")
            cols.append(f"{ta:7.1f} / {tp:7.1f} us ({fl / tp / 1e6:5.0f} TF)")
            if ta == ta:
                best = min(best, ta)
        force(-1)
        call("mdvit_gemm_ph_config", 1)
        print(f"{note:14s} {M:6d}x{N:5d}x{K:5d}  {t_old:7.1f} us {fl / t_old / 1e6:5.0f} TF | {best128[0]:7.1f} / {best128[1]:7.1f} us ({fl / best128[1] / 1e6:5.0f} TF) | "
              + " | ".join(cols) + f" | {fl / best / 1e6:5.0f}", flush=True)




def timing_epilogues():
    """the fused-epilogue launches of a C >= 320 block (fp32 A, weight planes): old split-while-staging kernel vs the 128 plane tiles vs the 256 tile"""
    print("\nepilogue timing (fp32 A):   gemm.hip | plane 128 tiles (best) | cfg 3", flush=True)
    cases = [("fc1 gelu dual s2", 32768, 1280, 320, "gelu"), ("fc2 dgrad dgelu s2", 32768, 1280, 320, "dgelu"), ("qkv plain s2", 32768, 960, 320, "plain"),
             ("fc2 full s2", 32768, 320, 1280, "full"), ("fc1 gelu dual s3", 8192, 2048, 512, "gelu"), ("fc2 dgrad dgelu s3", 8192, 2048, 512, "dgelu"),
             ("qkv plain s3", 8192, 1536, 512, "plain"), ("fc2 full s3", 8192, 512, 2048, "full"),
             ("fc1 gelu dual s2 bs4", 16384, 1280, 320, "gelu"), ("fc1 gelu dual s3 bs4", 4096, 2048, 512, "gelu"),
             ("fc1 gelu dual s2 step32", 131072, 1280, 320, "gelu"), ("fc2 dgrad dgelu s2 step32", 131072, 1280, 320, "dgelu"), ("qkv plain s2 step32", 131072, 960, 320, "plain"),
             ("fc1 gelu dual s3 step32", 32768, 2048, 512, "gelu"), ("fc2 dgrad dgelu s3 step32", 32768, 2048, 512, "dgelu"), ("fc2 full s3 step32", 32768, 512, 2048, "full"),
             ("proj full s3 step32", 32768, 512, 512, "full"), ("qkv plain s3 step32", 32768, 1536, 512, "plain"), ("fc1 dgrad plain s3 step32", 32768, 512, 2048, "plain")]
    for note, M, N, K, kind in cases:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
        wp = planes_of(w)
        out = torch.empty((M, N), device="cuda"); U = torch.empty((M, N), device="cuda"); res = torch.randn(M, N, device="cuda")
        rs = torch.rand((M + 1023) // 1024, device="cuda")

        def old():
            if kind == "gelu":
                ops.gemm(ops._p(x), ops._p(w), ops._p(U), M, N, K, lda=K, ldb=K, ldc=N, precision=1, bias=ops._p(b), out2=ops._p(out), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2))
            elif kind == "dgelu":
                ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1, epi=_lib.EPI_DGELU, gelu_u=ops._p(U), ldu=N, e_drop=0.1, e_key=(1, 2))
            elif kind == "full":
                ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1, bias=ops._p(b), residual=ops._p(res), ldr=N, e_drop=0.1, e_key=(1, 2),
                         e_rowscale=ops._p(rs), e_rows_per_scale=1024)
            else:
                ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1, bias=ops._p(b))

        def new():
            if kind == "gelu":
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, epi=_lib.EPI_GELU_DUAL, C_out=out, U=U, drop=0.1, key=(1, 2))
            elif kind == "dgelu":
                run_bp(x, wp, M, N, K, a_f32=True, epi=_lib.EPI_DGELU, C_out=out, gelu_u=U, drop=0.1, key=(1, 2))
            elif kind == "full":
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=out, residual=res, rowscale=rs, rps=1024, drop=0.1, key=(1, 2))
            else:
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=out)
        force(-1)
        t_old = time_it(old)
        t128 = 1e9
        for cfg in (0, 1, 2):
            force(cfg); t128 = min(t128, time_it(new))
        force(3); t3 = time_it(new); force(-1)
        fl = 2.0 * M * N * K
        print(f"{note:24s} {M:6d}x{N:5d}x{K:5d}  {t_old:7.1f} us | {t128:7.1f} us | {t3:7.1f} us ({fl / t3 / 1e6:5.0f} TF)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--cfgs", default="3")
    ap.add_argument("--variants", default="1")
    ap.add_argument("--epilogues", action="store_true")
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(",")]
    good = True
    variants = [int(v) for v in a.variants.split(",")]
    if not a.no_check:
        for v in variants:
            call("mdvit_gemm_ph_config", v)
            print(f"---- main-loop variant {v}", flush=True)
            good &= correctness(cfgs)
        call("mdvit_gemm_ph_config", 1)
        print("CORRECTNESS", "PASS" if good else "FAIL", flush=True)
    if not a.no_timing:
        timing(a.quick, cfgs, variants)
    if a.epilogues:
        timing_epilogues()
    sys.exit(0 if good else 1)
