"""Fused MLP kernels (C = 64): 64- vs 128-token tiles -- same results, timing.   python tools/mlp_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops
from mdvit_amd._lib import call


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
C, Hd = 64, 512
W1 = (torch.randn(Hd, C, device="cuda") * C ** -0.5).requires_grad_(True); b1 = (torch.randn(Hd, device="cuda") * 0.1).requires_grad_(True)
W2 = (torch.randn(C, Hd, device="cuda") * Hd ** -0.5).requires_grad_(True); b2 = (torch.randn(C, device="cuda") * 0.1).requires_grad_(True)
for M in (4096 + 77, 262144, 524288):
    x = torch.randn(M, C, device="cuda", requires_grad=True); res = torch.randn(M, C, device="cuda"); g = torch.randn(M, C, device="cuda")
    rs = (torch.rand(4, device="cuda") < 0.9).float() / 0.9
    outs = {}
    for wide in (0, 1):               # 0: 64-token tile, 1: 128-token tile
        call("mdvit_mlp_config", 1 if wide == 1 else -1, 0)
        import itertools
        ops._key_counter = itertools.count(5)

        def fwd():
            return ops.mlp_residual(x, res, W1, b1, W2, b2, rowscale=rs, drop_p=0.1, rows_per_scale=(M + 3) // 4)
        y = fwd()
        for t in (x, W1, b1, W2, b2):
            t.grad = None
        y.backward(g)
        outs[wide] = [y.detach().clone(), x.grad.clone(), W1.grad.clone(), W2.grad.clone()]
        with torch.no_grad():
            t_f = timed(fwd)

        def fb():
            yy = fwd(); yy.backward(g)
        t_fb = timed(fb, 5)
        print(f"M={M:7d} wide={wide}: fwd {t_f:8.1f} us   fwd+bwd {t_fb:8.1f} us", flush=True)
    for name, a, b in zip(("y", "dx", "dW1", "dW2"), outs[0], outs[1]):
        err = float((a - b).abs().max() / b.abs().max())
        print(f"   {name}: 64- vs 128-token tile max rel diff {err:.2e}")
# where the forward's time goes: switch parts of the kernel off (results wrong by design)
M = 262144
x = torch.randn(M, C, device="cuda"); res = torch.randn(M, C, device="cuda")
for wide in (0, 1):
    for abl, what in ((0, "full"), (1, "no h store"), (2, "no weight reloads"), (4, "no GELU"), (8, "no second product"), (3, "no h store, no weight reloads"), (15, "all four off")):
        call("mdvit_mlp_config", 1 if wide else -1, abl)
        with torch.no_grad():
            t = timed(lambda: ops.mlp_residual(x, res, W1, b1, W2, b2, rowscale=None, drop_p=0.1, rows_per_scale=M))
        print(f"ablation M={M} wide={wide} {what:32s} {t:8.1f} us", flush=True)
call("mdvit_mlp_config", 32768, 0)
