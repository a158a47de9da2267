"""fused MLP forward vs the two-GEMM form at the stage-0 shape (fused 16-image batch): us per call"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
M, C, Hd = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, 64, 512
dev = "cuda:0"
x, res = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
W1, b1 = torch.randn(Hd, C, device=dev) * 0.1, torch.randn(Hd, device=dev) * 0.1
W2, b2 = torch.randn(C, Hd, device=dev) * 0.05, torch.randn(C, device=dev) * 0.1
rs = (torch.rand(16, device=dev) < 0.9).float() / 0.9
def run():
    with torch.no_grad():
        return ops.mlp_residual(x, res, W1, b1, W2, b2, rowscale=rs, drop_p=0.1, rows_per_scale=M // 16)
for flag in (True, False, True, False):
    ops._mlp_fused = flag
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    by = 4.0 * (M * C * 3 + M * Hd * (1 if flag else 2))
    print(f"fused={flag}: {e0.elapsed_time(e1) * 100:.1f} us  ({by / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12:.2f} TB/s of the form's own traffic)")
