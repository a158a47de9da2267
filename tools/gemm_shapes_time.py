"""Every product of one bench step, timed ALONE on the GPU in the planner's own configuration and priced against its roofline:
    MDVIT_BENCH_GEMM_SHAPES=shapes.txt python bench.py --steps 3 --warmup 2 --no-extra-legs --no-cpu-baseline      (writes the step's product list: the library's launch ledger, mode 2)
    python tools/gemm_shapes_time.py shapes.txt [kernel-name filter]
Per row: launches per step, us alone, GB/s and TF/s, the larger of (bytes / 8 TB/s, flop / bf16x3 roof) over the measured time.  The last line weighs the rows by launches per step --
the isolated counterpart of bench.py's `roofline.frac` (which is measured inside the three-stream step).  The implicit 3x3 convolutions are listed but not timed here."""
import os
import re
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops

HBM, MFMA3 = 8.0e12, 2.5e15 / 3


def timed(fn, n=10):
    for _ in range(30):          # (the first ~30 launches after an idle gap run ~20 % slower: clock ramp)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    pat = re.compile(r"\s*(\d+) x\s+([\d.]+) MB\s+([\d.]+) GF\s+(\S+<[^>]*>) M=(\d+) N=(\d+) K=(\d+) ta=(\d) tb=(\d)(.*)")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    x = torch.empty(64 << 20, device="cuda")          # 256 MB written between timings would be cleaner; the products below are larger than the 4 MB L2 + 256 MB MALL only partly
    tot = {}
    for line in open(sys.argv[1]):
        m = pat.match(line)
        if not m:
            continue
        n, mb, gf, name, M, N, K, ta, tb, extra = m.groups()
        n, M, N, K, ta, tb = int(n), int(M), int(N), int(K), int(ta), int(tb)
        if "conv3x3" in name or flt not in name:
            continue
        by, fl = float(mb) * 1e6, float(gf) * 1e9
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((N, K) if tb else (K, N), device="cuda")
        out = torch.empty((M, N), device="cuda")
        kw = dict(allow_split=True)
        if "+u" in extra:
            u = torch.randn((M, N), device="cuda")
            kw = dict(epi=_lib.EPI_DGELU, gelu_u=ops._p(u), ldu=N, e_drop=0.1, e_key=(1, 2))
        elif "+C2" in extra:
            bias, out2 = torch.randn(N, device="cuda"), torch.empty_like(out)
            kw = dict(bias=ops._p(bias), epi=_lib.EPI_GELU_DUAL, e_drop=0.1, e_key=(1, 2), out2=ops._p(out2))
        elif "+res" in extra:
            res, bias = torch.randn((M, N), device="cuda"), torch.randn(N, device="cuda")
            kw = dict(residual=ops._p(res), ldr=N, bias=ops._p(bias), e_drop=0.1, e_key=(1, 2))
        if ta != tb:
            kw["precision"] = 1

        def run():
            ops.gemm(ops._p(A), ops._p(B), ops._p(out), M, N, K, lda=A.stride(0), ldb=B.stride(0), ldc=N, trans_a=bool(ta), trans_b=bool(tb), **kw)
        t = timed(run)
        bound = max(by / HBM, fl / MFMA3) * 1e6
        key = name.split("<")[0] + ("" if ta == 0 else " (TN)")
        a = tot.setdefault(key, [0.0, 0.0, 0])
        a[0] += n * t; a[1] += n * bound; a[2] += n
        print(f"{n:3d} x {t:8.1f} us  {by / t / 1e3:7.0f} GB/s {fl / t / 1e6:6.1f} TF/s  bound {bound:6.1f} us  frac {bound / t:5.2f}   {name} M={M} N={N} K={K}{extra}", flush=True)
        del A, B, out
    for k, (t, b, n) in tot.items():
        print(f"== {k}: {n} launches per step, {t / 1e3:.2f} ms alone, bound {b / 1e3:.2f} ms, frac {b / t:.3f}")


main()
