"""fused MLP forward + backward launches at the stage-0 shape for a rocprofv3 --pmc pass"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
M, C, Hd = 262144, 64, 512
dev = "cuda:0"
x = torch.randn(M, C, device=dev, requires_grad=True); res = torch.randn(M, C, device=dev)
W1 = torch.nn.Parameter(torch.randn(Hd, C, device=dev) * 0.1); b1 = torch.nn.Parameter(torch.randn(Hd, device=dev) * 0.1)
W2 = torch.nn.Parameter(torch.randn(C, Hd, device=dev) * 0.05); b2 = torch.nn.Parameter(torch.randn(C, device=dev) * 0.1)
g = torch.randn(M, C, device=dev)
for _ in range(3):
    y = ops.mlp_residual(x, res, W1, b1, W2, b2, drop_p=0.1)
    ops.set_dgrad_only(True)
    y.backward(g)
    ops.set_dgrad_only(False)
torch.cuda.synchronize()
