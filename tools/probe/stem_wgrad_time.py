"""stem.0 weight gradient alone (mdvit_stemconv_wgrad: 16 images 3 x 512 x 512 -> 32 channels, the bs=4 step's shape): MDVIT_STEM_WGRAD_MFMA=0 python tools/probe/stem_wgrad_time.py for the scalar kernel"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
from mdvit_amd._lib import call

B, H, W, Cout = 16, 512, 512, 32
img = torch.randn(B, 3, H, W, device="cuda")
g = torch.randn(B, H // 2, W // 2, Cout, device="cuda")
dw = torch.empty(Cout, 3, 3, 3, device="cuda")
wsp, wsb, keep = ops._partials_ws(27 * Cout, img.device)
fn = lambda: call("mdvit_stemconv_wgrad", ops._p(img), ops._p(g), ops._p(dw), wsp, wsb, B, H, W, 3, Cout, 0, ops._stream())
for _ in range(30):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record(); torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(img.double(), torch.zeros(Cout, 3, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True), None, 2, 1)
w0 = torch.zeros(Cout, 3, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
torch.nn.functional.conv2d(img.double(), w0, None, 2, 1).backward(g.permute(0, 3, 1, 2).double())
err = float((dw.double() - w0.grad).abs().max() / w0.grad.abs().max())
print(f"MDVIT_STEM_WGRAD_MFMA={os.environ.get('MDVIT_STEM_WGRAD_MFMA', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (kernel + partial-row reduction), max rel err vs fp64 {err:.2e}")
