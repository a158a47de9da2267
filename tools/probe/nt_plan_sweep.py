"""Every NT / NN product of one bench step (the ledger file of MDVIT_BENCH_GEMM_SHAPES, see tools/gemm_shapes_time.py) under FORCED gemm.hip plans against the planner's own choice:
where does the cost model pick a plan that loses by more than 10 %?   python tools/probe/nt_plan_sweep.py profiles/r05_gemm_shapes_bs4.txt"""
import os, re, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import _lib, ops
from mdvit_amd._lib import call


def timed(fn, n=10):
    for _ in range(25):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pat = re.compile(r"\s*(\d+) x\s+([\d.]+) MB\s+([\d.]+) GF\s+(\S+<[^>]*>) M=(\d+) N=(\d+) K=(\d+) ta=(\d) tb=(\d)(.*)")
tot_pl = tot_best = 0.0
for line in open(sys.argv[1]):
    m = pat.match(line)
    if not m:
        continue
    n, mb, gf, name, M, N, K, ta, tb, extra = m.groups()
    n, M, N, K, ta, tb = int(n), int(M), int(N), int(K), int(ta), int(tb)
    if "conv3x3" in name or ta or not name.startswith("gemm_f32_kernel"):
        continue
    A = torch.randn((M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty((M, N), device="cuda")
    kw = dict(allow_split=True)
    plain = True
    if "+u" in extra:
        u = torch.randn((M, N), device="cuda"); plain = False
        kw = dict(epi=_lib.EPI_DGELU, gelu_u=ops._p(u), ldu=N, e_drop=0.1, e_key=(1, 2))
    elif "+C2" in extra:
        bias, out2 = torch.randn(N, device="cuda"), torch.empty_like(out); plain = False
        kw = dict(bias=ops._p(bias), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2), out2=ops._p(out2))
    elif "+res" in extra:
        res, bias = torch.randn((M, N), device="cuda"), torch.randn(N, device="cuda"); plain = False
        kw = dict(residual=ops._p(res), ldr=N, bias=ops._p(bias), e_drop=0.1, e_key=(1, 2))
    if tb:
        kw["precision"] = 1

    def run():
        ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=K, ldb=B.stride(0), ldc=N, trans_a=False, trans_b=bool(tb), **kw)
    t_pl = timed(run)
    res_ = []
    for cfg in (0, 1, 2):
        for sp in ((1, 2, 3, 4, 6, 8) if plain and K >= 512 else (1,)):
            call("mdvit_gemm_force_plan", cfg, sp)
            try:
                res_.append((timed(run, 6), cfg, sp))
            except Exception:
                pass
            finally:
                call("mdvit_gemm_force_plan", -1, 0)
    t_b, c_b, s_b = min(res_)
    tot_pl += n * t_pl; tot_best += n * min(t_b, t_pl)
    flag = "  <-- planner loses" if t_b < 0.9 * t_pl else ""
    print(f"{n:3d} x M={M:7d} N={N:5d} K={K:5d}{extra:12s} {name[16:44]:28s} planner {t_pl:7.1f} us | best forced cfg {c_b} sp {s_b}: {t_b:7.1f} us{flag}", flush=True)
    del A, B, out
print(f"== per step: planner {tot_pl / 1e3:.2f} ms, best forced {tot_best / 1e3:.2f} ms")
