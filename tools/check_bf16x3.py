"""Measure the error of the bf16x3 GEMM paths against fp64 (rel to tensor max and rel L2)."""
import torch
from mdvit_amd import ops

torch.manual_seed(0)
def err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())

for (M, N, K) in [(512, 512, 64), (512, 64, 512), (4096, 320, 1280), (65536, 64, 64)]:
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; g = torch.randn(M, N, device="cuda")
    ref = x.double() @ W.double().t(); dxr = g.double() @ W.double(); dwr = g.double().t() @ x.double()
    for prec in ("fp32", "bf16x3"):
        ops.set_gemm_precision(prec)
        xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
        y = ops.linear(xr, Wr, None)
        y.backward(g)
        print(f"M={M} N={N} K={K} {prec:7s} y {err(y, ref)}  dx {err(xr.grad, dxr)}  dW {err(Wr.grad, dwr)}")
