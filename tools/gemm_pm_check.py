"""The 128-row phase-split plane GEMM (csrc/gemm_pm.hip; mdvit_gemm_planes with the plan forced to cfg 6 = 128 x 160 / 7 = 128 x 128) on the GPU:
BIT equality with the 128 x 128 plane tile for every epilogue (plain, + accumulate, GELU dual, DGELU, full) on shapes with ragged edges, a race screen
(20 repeats must agree bit for bit), fp64 error, and timing against gemm.hip (ops.gemm, what the step ran before) on the mid-size shapes of the step.
    python tools/gemm_pm_check.py [--no-timing]"""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mdvit_amd import _lib, ops  # noqa: E402
from mdvit_amd._lib import call  # noqa: E402
from gemm_bp_check import planes_of, run_bp, gelu_grad, check  # noqa: E402


def force(cfg):
    call("mdvit_gemm_planes_force_plan", cfg, 0)


def correctness():
    torch.manual_seed(0)
    ok = True
    for cfg, shapes in ((6, ((512, 320, 64), (300, 160, 128), (1000, 320, 320), (257, 960, 96), (2048, 1280, 320), (4096, 320, 1280), (130, 164, 32))),
                        (7, ((512, 512, 64), (300, 128, 128), (1000, 384, 320), (257, 260, 1024), (4096, 512, 2048)))):
        for (M, N, K) in shapes:
            x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
            b = torch.randn(N, device="cuda")
            x += torch.arange(M, device="cuda", dtype=torch.float32)[:, None] * 1e-3          # asymmetric, transpose-detecting data
            w += torch.arange(N, device="cuda", dtype=torch.float32)[:, None] * 1e-4
            ref = x.double() @ w.double().T + b.double()
            wp = planes_of(w)
            res = torch.randn(M, N, device="cuda"); rs = torch.rand((M + 49) // 50, device="cuda"); u = torch.randn(M, N, device="cuda")
            acc0 = torch.randn(M, N, device="cuda")

            def all_outputs():
                outs = {}
                o = torch.full((M, N), float("nan"), device="cuda")
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=o); outs["plain"] = o
                o = acc0.clone()
                run_bp(x, wp, M, N, K, a_f32=True, C_out=o, accumulate=True); outs["accumulate"] = o
                U = torch.full((M, N), float("nan"), device="cuda"); h = torch.full((M, N), float("nan"), device="cuda")
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, epi=_lib.EPI_GELU_DUAL, C_out=h, U=U, drop=0.1, key=(3, 4)); outs["gelu_u"] = U; outs["gelu_h"] = h
                o = torch.full((M, N), float("nan"), device="cuda")
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=o, residual=res, rowscale=rs, rps=50, drop=0.1, key=(5, 6)); outs["full"] = o
                o = torch.full((M, N), float("nan"), device="cuda")
                run_bp(x, wp, M, N, K, a_f32=True, epi=_lib.EPI_DGELU, C_out=o, gelu_u=u, drop=0.1, key=(7, 8)); outs["dgelu"] = o
                return outs
            force(0)
            base = all_outputs()
            force(cfg)
            got = all_outputs()
            for k in base:
                same = bool(torch.equal(got[k], base[k]))
                print(f"  {'ok ' if same else 'BAD'} cfg {cfg} {M}x{N}x{K} {k:10s} == the 128x128 plane tile bit for bit: {same}"
                      + ("" if same else f"  (max abs diff {float((got[k] - base[k]).abs().nan_to_num(1e9).max()):.3e})"), flush=True)
                ok &= same
            ok &= check(f"cfg {cfg} {M}x{N}x{K} plain vs fp64", got["plain"], ref, 2e-5)
            ok &= check(f"cfg {cfg} {M}x{N}x{K} dgelu vs fp64 (no drop)", _nodrop_dgelu(x, wp, u, M, N, K), (x.double() @ w.double().T) * gelu_grad(u.double()), 2e-5)
            rep_ok = True
            for _ in range(20):
                o2 = torch.full((M, N), float("nan"), device="cuda")
                run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=o2)
                rep_ok &= bool(torch.equal(o2, got["plain"]))
            print(f"  {'ok ' if rep_ok else 'BAD'} cfg {cfg} {M}x{N}x{K} 20 repeats identical: {rep_ok}", flush=True)
            ok &= rep_ok
    force(-1)
    return ok


def _nodrop_dgelu(x, wp, u, M, N, K):
    o = torch.empty((M, N), device="cuda")
    run_bp(x, wp, M, N, K, a_f32=True, epi=_lib.EPI_DGELU, C_out=o, gelu_u=u)
    return o


def time_it(fn, iters=20):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [
    (16384, 320, 320, 6, "proj s2 bs4"), (16384, 320, 960, 6, "qkv dgrad s2 bs4"), (16384, 320, 1280, 6, "fc2 s2 bs4"), (16384, 1280, 320, 6, "fc1 s2 bs4"),
    (16384, 960, 320, 6, "qkv s2 bs4"),
    (32768, 320, 320, 6, "proj s2 bs32"), (32768, 320, 1280, 6, "fc2 s2 bs32"), (32768, 1280, 320, 6, "fc1 s2 bs32"), (32768, 960, 320, 6, "qkv s2 bs32"),
    (4096, 512, 512, 7, "proj s3 bs4"), (4096, 512, 2048, 7, "fc2 s3 bs4"), (4096, 2048, 512, 7, "fc1 s3 bs4"), (4096, 1536, 512, 7, "qkv s3 bs4"),
    (8192, 512, 512, 7, "proj s3 bs32"), (8192, 512, 2048, 7, "fc2 s3 bs32"), (8192, 2048, 512, 7, "fc1 s3 bs32"), (8192, 1536, 512, 7, "qkv s3 bs32"),
    (131072, 320, 1280, 6, "fc2 s2 step32"), (32768, 512, 2048, 7, "fc2 s3 step32"),
]


def timing():
    print("\nshape (plain epilogue, fp32 A)              gemm.hip | planes, planner | 128-row phase-split tile | with the full (residual) epilogue: gemm.hip | pm", flush=True)
    for (M, N, K, cfg, note) in SHAPES:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
        out = torch.empty((M, N), device="cuda"); res = torch.randn(M, N, device="cuda"); b = torch.randn(N, device="cuda")
        wp = planes_of(w)
        fl = 2.0 * M * N * K
        force(-1)
        t_old = time_it(lambda: ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1))
        call("mdvit_gemm_pm_config", -1)
        t_pl = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
        call("mdvit_gemm_pm_config", 0)
        force(cfg)
        t_pm = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
        t_pm_full = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, bias=b, C_out=out, residual=res, drop=0.1, key=(5, 6)))
        force(-1)
        t_old_full = time_it(lambda: ops.gemm(ops._p(x), ops._p(w), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1, residual=ops._p(res), ldr=N, bias=ops._p(b),
                                              e_drop=0.1, e_key=(5, 6)))
        takes = _lib.load().mdvit_gemm_pm_prefers(M, N, K, 2, 1)
        print(f"{note:18s} {M:6d}x{N:5d}x{K:5d}  {t_old:7.1f} us {fl / t_old / 1e6:5.0f} TF | {t_pl:7.1f} us | cfg {cfg}: {t_pm:7.1f} us {fl / t_pm / 1e6:5.0f} TF "
              f"| {t_old_full:7.1f} | {t_pm_full:7.1f} us | rule takes it: {takes}", flush=True)


def timing_splits():
    """the few-tile long-K data gradients of stage 3 at 16 images: gemm.hip (what ran before) against the 128-row tile over the rule's K ranges (+ the reduction launch)"""
    print("\nplain products with few tiles and a long K: gemm.hip (planner) | 128-row tile unsplit | over the rule's K ranges (+ reduce) | the 128 x 128 plane tile, same ranges", flush=True)
    for (M, N, K, note) in ((4096, 512, 2048, "fc1^T dgrad s3 bs4"), (4096, 512, 1536, "qkv^T dgrad s3 bs4"), (2048, 512, 2048, "fc1^T dgrad s3 bs2"), (4096, 320, 1280, "fc1^T dgrad s2 bs1")):
        g = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.1
        Wt = W.t().contiguous()
        out = torch.empty((M, N), device="cuda")
        sp_rule = _lib.load().mdvit_gemm_pm_splits(M, N, K, 2, 1)
        sp = sp_rule if sp_rule > 1 else 2          # (where the rule declines: what two ranges would cost)
        fl = 2.0 * M * N * K
        force(-1)
        t_old = time_it(lambda: ops.gemm(ops._p(g), ops._p(Wt), ops._p(out), M, N, K, lda=K, ldb=K, ldc=N, precision=1, allow_split=True))
        cfg = 6 if N % 160 == 0 else 7
        call("mdvit_gemm_planes_force_plan", cfg, 0)
        t_un = time_it(lambda: ops.gemm_nt(g, W, out, M, N, K, w_transposed=True, allow_split=True))
        call("mdvit_gemm_planes_force_plan", cfg, sp)
        t_sp = time_it(lambda: ops.gemm_nt(g, W, out, M, N, K, w_transposed=True, allow_split=True))
        call("mdvit_gemm_planes_force_plan", 0, sp)
        t_bp = time_it(lambda: ops.gemm_nt(g, W, out, M, N, K, w_transposed=True, allow_split=True))
        force(-1)
        print(f"{note:20s} {M:6d}x{N:5d}x{K:5d}  {t_old:7.1f} us {fl / t_old / 1e6:5.0f} TF | {t_un:7.1f} us | {sp} ranges: {t_sp:7.1f} us {fl / t_sp / 1e6:5.0f} TF | {t_bp:7.1f} us | rule: {sp_rule}", flush=True)


def timing_presplit():
    """VERDICT r05 item 8 -- "producer-written hi / lo planes" (Next (1)) priced with what exists: on the nine mid-size shapes, the A operand as fp32
    (split while staged: what the step runs) against A ALREADY split into bf16 planes by its producer (half the A bytes per plane pair, no split VALU in the main loop),
    on every tile that takes plane A (the 128 x 128 / 64-row tiles of gemm_bp.hip and the 256-wide phase-split tile; gemm_pm.hip takes fp32 A only)."""
    print("\nA fp32 (split while staged) against A pre-split into planes -- shape: gemm_pm fp32 A | 128x128 plane tile fp32 A | same tile, plane A | planner, plane A | best plane-A / gemm_pm", flush=True)
    for (M, N, K, cfg, note) in SHAPES[:17]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1
        out = torch.empty((M, N), device="cuda")
        wp = planes_of(w); xp = planes_of(x)
        call("mdvit_gemm_pm_config", 0)
        force(cfg)
        t_pm = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
        force(0)
        t_bp_f32 = time_it(lambda: run_bp(x, wp, M, N, K, a_f32=True, C_out=out))
        t_bp_pl = time_it(lambda: run_bp(xp, wp, M, N, K, a_f32=False, C_out=out))
        force(-1)
        t_auto_pl = time_it(lambda: run_bp(xp, wp, M, N, K, a_f32=False, C_out=out))
        best = min(t_bp_pl, t_auto_pl)
        print(f"{note:18s} {M:6d}x{N:5d}x{K:5d}  {t_pm:7.1f} us | {t_bp_f32:7.1f} us | {t_bp_pl:7.1f} us | {t_auto_pl:7.1f} us | pm / best plane-A = {t_pm / best:4.2f}x", flush=True)


if __name__ == "__main__":
    if "--presplit" in sys.argv:
        timing_presplit()
        sys.exit(0)
    good = correctness()
    print("CORRECTNESS", "OK" if good else "FAILED", flush=True)
    if "--no-timing" not in sys.argv:
        timing()
        timing_splits()
    sys.exit(0 if good else 1)
