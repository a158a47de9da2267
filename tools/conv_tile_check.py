"""Stencil tiles: two rows per thread on packed FMAs (MDVIT_CONV_TILE2, default) against one row per thread -- timing of the attention's ConvRelPosEnc
passes and the ConvPosEnc depthwise conv at the four stage shapes.  Run once per setting:  MDVIT_CONV_TILE2=0|1 python tools/conv_tile_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (H, C, heads) in ((128, 64, 8), (64, 128, 8), (32, 320, 8), (16, 512, 8)):
    x = torch.randn(B, H, H, C, device="cuda", requires_grad=True)
    w = torch.randn(C, 1, 3, 3, device="cuda") * 0.3; b = torch.randn(C, device="cuda") * 0.1
    y = ops.dwconv3x3(x, w, b, 1, True)
    g = torch.randn_like(y)
    t_f = timed(lambda: ops.dwconv3x3(x.detach(), w, b, 1, True))
    t_fb = timed(lambda: torch.autograd.grad(ops.dwconv3x3(x, w, b, 1, True), x, g))
    print(f"H={H:3d} C={C:3d}: dwconv3x3(+x) fwd {t_f:7.1f} us   fwd + dgrad {t_fb:7.1f} us   checksum {float(y.double().sum()):.6f}", flush=True)
