"""The weight-gradient kernel with operands that never leave the caches (lda = ldb = 4: the token rows overlap, ~1 MB in all) against the same product from HBM: how much of its time is memory?   python tools/probe/tn_cached_operands_probe.py"""
import os, sys, torch
_r = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _r); sys.path.insert(0, os.path.join(_r, "tools"))
from mdvit_amd import _lib, ops
from tn_check import timed, run
lib = _lib.load()
for (M, N, K) in ((1024, 128, 65536), (1280, 320, 16384), (64, 512, 262144)):
    A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); out = torch.zeros((M, N), device="cuda")
    t_real = timed(lambda: run(A, B, out, M, N, K, accumulate=True))
    try:
        t_l2 = timed(lambda: run(A, B, out, M, N, K, lda=4, ldb=4, accumulate=True))
    except Exception as e:
        t_l2 = float("nan"); print("lda=4 rejected:", str(e)[:100])
    # a K range that fits the L2 / infinity cache: rows wrap every 256 tokens
    print(f"M={M} N={N} K={K}: HBM operands {t_real:.1f} us, rows 16 bytes apart (lda = ldb = 4: the operands are ~1 MB, all loads hit the caches) {t_l2:.1f} us", flush=True)
