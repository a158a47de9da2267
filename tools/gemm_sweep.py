"""Ground truth for the GEMM planner: time every (tile config, K-split) of the shapes a bench step uses and compare
with the planner's pick.  usage: python tools/gemm_sweep.py shapes.json [top]   (shapes.json from bench.py --by-shape --detail)"""
import ctypes as C, json, re, sys
import torch
from mdvit_amd import _lib, ops

lib = _lib.load()
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256]
CFG = {0: "128x128", 1: "256x64", 2: "64x64"}


def timed(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    d = json.load(open(sys.argv[1]))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    seen, total_pl, total_best = set(), 0.0, 0.0
    for name, r in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:top]:
        m = re.search(r"<\d+, \d+, \d, \d, (\w+), (\w+), (\d)(?:, \w+)?>.* M=(\d+) N=(\d+) K=(\d+)", name)
        ta, tb, epi, M, N, K = m.groups()
        ta, tb, M, N, K = ta == "true", tb == "true", int(M), int(N), int(K)
        epi = int(epi)
        if (M, N, K, ta, tb, epi) in seen:
            continue
        seen.add((M, N, K, ta, tb, epi))
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((N, K) if tb else (K, N), device="cuda")
        out = torch.empty((M, N), device="cuda")

        extra = {}
        if epi == 4:
            continue                         # the recomputing DGELU is built on one tile shape only: nothing to plan
        if epi == 1:
            bias = torch.randn(N, device="cuda")
            extra = dict(bias=ops._p(bias), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2))
            if K > 128:                      # the wide stages store u and gelu(u); the C <= 128 ones gelu(u) only
                out2 = torch.empty_like(out)
                extra["out2"] = ops._p(out2)
        elif epi == 2:
            u = torch.randn((M, N), device="cuda")
            extra = dict(epi=_lib.EPI_DGELU, gelu_u=ops._p(u), ldu=N, e_drop=0.1, e_key=(1, 2))
        elif epi == 3:
            res = torch.randn((M, N), device="cuda"); bias = torch.randn(N, device="cuda")
            extra = dict(residual=ops._p(res), ldr=N, bias=ops._p(bias), e_drop=0.1, e_key=(1, 2))

        def run():
            ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=A.stride(0), ldb=B.stride(0), ldc=N, trans_a=ta, trans_b=tb,
                     allow_split=epi == 0, **extra)
        if ta != tb:
            extra["precision"] = 1          # the step runs these layouts in bf16x3
        lib.mdvit_gemm_force_plan(-1, 0)
        t_pl = timed(run)
        g = _lib.GemmDesc(); g.M, g.N, g.K, g.trans_a, g.trans_b, g.allow_split = M, N, K, int(ta), int(tb), int(epi == 0)
        g.precision = extra.get("precision", 0)
        g.epi = {0: 0, 1: _lib.EPI_GELU_DUAL, 2: _lib.EPI_DGELU, 3: 0}[epi]
        if epi == 3:
            g.residual = 1
        tm, tn, sp = C.c_int32(), C.c_int32(), C.c_int32()
        lib.mdvit_gemm_plan(C.byref(g), C.byref(tm), C.byref(tn), C.byref(sp))
        res = []
        for cfg in range(3):
            for s in SPLITS:
                if s > 1 and (epi != 0 or K < 512 or s > K // 256):
                    break
                lib.mdvit_gemm_force_plan(cfg, s)
                res.append((timed(run, 4), cfg, s))
        lib.mdvit_gemm_force_plan(-1, 0)
        res.sort()
        bt, bc, bs = res[0]
        total_pl += t_pl * r["n"]; total_best += bt * r["n"]
        t_bf = None
        fl = 2.0 * M * N * K
        print(f"M={M:>7} N={N:>5} K={K:>7} {'T' if ta else 'N'}{'T' if tb else 'N'} e{epi} x{r['n']:3d}  planner {tm.value}x{tn.value} sp={sp.value:<4d} {t_pl:8.1f} us ({fl/t_pl/1e6:5.1f} TF)"
              f" | best {CFG[bc]} sp={bs:<4d} {bt:8.1f} us ({fl/bt/1e6:5.1f} TF) | next {CFG[res[1][1]]} sp={res[1][2]} {res[1][0]:.1f}"
              + ("" if t_bf is None else f" | bf16x3 planner {t_bf:.1f} us ({fl/t_bf/1e6:.1f} TF) per cfg " + " ".join(f"{CFG[c]}:{t:.1f}" for t, c in bres)), flush=True)
    print(f"weighted: planner {total_pl/1e3:.2f} ms, best {total_best/1e3:.2f} ms")


main()
