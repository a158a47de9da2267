import os, sys
os.environ["MDVIT_POISON"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from mdvit_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for (M, K, N) in [(512, 64, 512), (512, 128, 512), (2048, 64, 512), (512, 64, 256), (512, 512, 64)]:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    Wf = torch.randn(N, N, device=dev, requires_grad=True)
    Wq = torch.randn(N, K, device=dev, requires_grad=True)
    Wc = ops.matmul(Wf, Wq)
    res = torch.randn(M, N, device=dev, requires_grad=True)
    y = ops.linear(x, Wc, None, residual=res)
    y.backward(torch.randn_like(y))
    print((M, K, N), "dx NaN", bool(torch.isnan(x.grad).any()), "dWf NaN", bool(torch.isnan(Wf.grad).any()), "dWq NaN", bool(torch.isnan(Wq.grad).any()))
    W = torch.nn.Parameter(torch.randn(N, K, device=dev))
    x2 = torch.randn(M, K, device=dev, requires_grad=True)
    ops.linear(x2, W, None).backward(torch.randn(M, N, device=dev))
    print("   leaf W: dx NaN", bool(torch.isnan(x2.grad).any()), "dW NaN", bool(torch.isnan(W.grad).any()))
