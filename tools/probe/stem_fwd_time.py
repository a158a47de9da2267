"""stem.0 forward alone (mdvit_stemconv_fwd, 16 x 3 x 512 x 512 -> 32 channels) and the per-step conv-weight layout refresh of an MDViT; MDVIT_STEM_FWD32=0 for the channel-quad kernel"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
from mdvit_amd._lib import call


def timed(fn, n=20):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, H, W, Cout = 16, 512, 512, 32
img = torch.randn(B, 3, H, W, device="cuda"); w = torch.randn(Cout, 3, 3, 3, device="cuda") * 0.2
y = torch.empty(B, H // 2, W // 2, Cout, device="cuda")
t = timed(lambda: call("mdvit_stemconv_fwd", ops._p(img), ops._p(w), ops._p(y), B, H, W, 3, Cout, ops._stream()))
ref = torch.nn.functional.conv2d(img[:2].double(), w.double(), None, 2, 1).permute(0, 2, 3, 1)
print(f"MDVIT_STEM_FWD32={os.environ.get('MDVIT_STEM_FWD32', '1')}: stem.0 forward {t:.1f} us, max abs err vs fp64 {float((y[:2].double() - ref).abs().max()):.2e}")
# the layouts of the dense 3x3 convolutions of an MDViT step (bridge 512 -> 512, 1024 -> 512; decoder / head convolutions), both modes
shapes = [(512, 512), (512, 1024), (512, 512), (320, 320), (128, 128), (64, 64), (256, 512), (64, 64), (32, 64)]
rows, keep = [], []
for co, ci in shapes:
    wt = torch.randn(co, ci, 3, 3, device="cuda")
    for mode in (0, 1):
        o = torch.empty(wt.numel(), device="cuda"); keep += [wt, o]
        rows.append([wt.data_ptr(), o.data_ptr(), co, ci, mode])
table = torch.tensor(rows, dtype=torch.int64, device="cuda")
mb = sum(r[2] * r[3] * 9 * 8 for r in rows) / 1e6
for blocks in (64, 256):
    t = timed(lambda: call("mdvit_conv_weight_relayout_many", ops._p(table), len(rows), blocks, ops._stream()))
    print(f"conv weight layouts, {len(rows)} items, {mb:.0f} MB moved, {blocks} workgroups per item: {t:.1f} us = {mb / t:.2f} TB/s")
