"""Compare two files written by `tools/mlp_rc_time.py --save`: per tensor, bit-for-bit equality or the largest relative difference.  python tools/compare_saved.py a.pt b.pt"""
import sys
import torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    x, y = a[k], b[k]
    same = torch.equal(x, y)
    d = float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
    print(f"{k:24s} {'bit-identical' if same else f'max |diff| / max |ref| = {d:.3e}'}")
