"""The bridge's dense 3x3 convolutions (512 -> 512 and 512 -> 1024 on 16 x 16 maps, 16 / 128 images) under forced GEMM plans: does the planner's choice (one K range) stand, or do 128-tile
launches want K ranges?   python tools/probe/conv_bridge_plans.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
from mdvit_amd._lib import call


def timed(fn, n=10):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
for B in (16, 128):
    for (Cin, Cout) in ((512, 512), (512, 1024), (1024, 512)):
        x = torch.randn(B, 16, 16, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
        b = torch.randn(Cout, device="cuda")
        with torch.no_grad():
            ref = ops.conv3x3_dense(x, w, b)
            row = []
            for cfg, sp in ((-1, 0), (0, 1), (0, 2), (0, 3), (0, 4), (2, 1), (2, 2), (1, 1), (1, 2)):
                call("mdvit_gemm_force_plan", cfg, sp)
                try:
                    y = ops.conv3x3_dense(x, w, b)
                    ok = float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
                    t = timed(lambda: ops.conv3x3_dense(x, w, b))
                    row.append(f"cfg {cfg} sp {sp}: {t:6.1f}{'' if ok else ' BAD'}")
                except Exception as e:
                    row.append(f"cfg {cfg} sp {sp}: {type(e).__name__}")
                finally:
                    call("mdvit_gemm_force_plan", -1, 0)
        fl = 2.0 * B * 256 * Cout * Cin * 9
        print(f"B={B:3d} {Cin}->{Cout} ({fl / 1e9:.1f} GF): " + " | ".join(row), flush=True)
