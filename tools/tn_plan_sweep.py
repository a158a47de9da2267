"""gemm_tn.hip planner sweep: for the weight-gradient shapes of the model, time every tile config x a range of TOTAL workgroup counts
(the K-split that gives it), next to the planner's own choice.   python tools/tn_plan_sweep.py [--bs32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops
from tn_check import timed, run

lib = _lib.load()
BM = (128, 128, 64, 64); BN = (128, 64, 128, 64)
tok = (262144, 65536, 16384, 4096)
if "--bs32" in sys.argv:
    tok = tuple(8 * t for t in tok)
shapes = [(64, 64, tok[0]), (192, 64, tok[0]), (64, 512, tok[0]), (512, 64, tok[0]), (128, 128, tok[1]), (384, 128, tok[1]), (1024, 128, tok[1]), (128, 1024, tok[1]),
          (320, 320, tok[2]), (960, 320, tok[2]), (1280, 320, tok[2]), (320, 1280, tok[2]), (512, 512, tok[3]), (1536, 512, tok[3]), (2048, 512, tok[3]), (512, 2048, tok[3])]
targets = (128, 192, 256, 320, 384, 512, 768, 1024)
for (M, N, K) in shapes:
    A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda"); out = torch.zeros((M, N), device="cuda")
    lib.mdvit_gemm_tn_config(1, -1, 0)
    t_plan = timed(lambda: run(A, B, out, M, N, K, accumulate=True))
    print(f"M={M:5d} N={N:5d} K={K:7d}: planner {t_plan:7.1f} us", flush=True)
    for cfg in range(4):
        tiles = -(-M // BM[cfg]) * -(-N // BN[cfg])
        row = []
        seen = set()
        for w in targets:
            sp = max(1, w // tiles)
            if sp > K // 256 or sp in seen:
                continue
            seen.add(sp)
            lib.mdvit_gemm_tn_config(1, cfg, sp)
            row.append(f"{tiles * sp:4d}wg(sp{sp:3d}) {timed(lambda: run(A, B, out, M, N, K, accumulate=True), 6):6.1f}")
        print(f"     cfg{cfg} {BM[cfg]:3d}x{BN[cfg]:3d} tiles {tiles:3d}: " + "  ".join(row), flush=True)
lib.mdvit_gemm_tn_config(1, -1, 0)
