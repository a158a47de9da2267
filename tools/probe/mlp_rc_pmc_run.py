"""mlp_rc forward + full backward launches at the stage-0 shape of bs=32 for a rocprofv3 --pmc / --kernel-trace pass"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
M, C, Hd = int(os.environ.get("M", 524288)), 64, 512
dev = "cuda:0"
x = torch.randn(M, C, device=dev, requires_grad=True); res = torch.randn(M, C, device=dev)
W1 = torch.nn.Parameter(torch.randn(Hd, C, device=dev) * 0.1); b1 = torch.nn.Parameter(torch.randn(Hd, device=dev) * 0.1)
W2 = torch.nn.Parameter(torch.randn(C, Hd, device=dev) * 0.05); b2 = torch.nn.Parameter(torch.randn(C, device=dev) * 0.1)
g = torch.randn(M, C, device=dev)
for _ in range(3):
    y = ops.mlp_residual(x, res, W1, b1, W2, b2, drop_p=float(os.environ.get("DROP", 0.1)))
    y.backward(g)
torch.cuda.synchronize()
