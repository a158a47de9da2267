import torch, time, os, sys
sys.path.insert(0, os.getcwd())
from mdvit_amd import ops
torch.manual_seed(0)
x = torch.randn(16, 256, 256, 32, device="cuda", requires_grad=True); w = (torch.randn(64, 32, 3, 3, device="cuda") * 0.05).requires_grad_()
g = torch.randn(16, 128, 128, 64, device="cuda")
for mode in ("0", "1"):
    os.environ["MDVIT_CONV_PHASE"] = mode
    for it in range(3):
        y = ops.conv3x3_dense(x, w, None, 2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y.backward(g); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("phase", mode, "backward (dgrad + wgrad) ms", dt * 1e3)
