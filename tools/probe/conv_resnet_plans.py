"""TransFuse's ResNet-34 3x3 convolutions (32 images of 256 x 256: 64 ch on 64 x 64, 128 on 32 x 32, 256 on 16 x 16) under forced GEMM plans against the planner's choice.
python tools/probe/conv_resnet_plans.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdvit_amd import ops
from mdvit_amd._lib import call


def timed(fn, n=10):
    for _ in range(25):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
warm = torch.randn(4096, 4096, device="cuda")
for _ in range(50):
    warm @ warm
for (B, S, Cin, Cout) in ((32, 64, 64, 64), (32, 32, 128, 128), (32, 16, 256, 256), (32, 32, 64, 128), (32, 16, 128, 256), (32, 64, 128, 64), (32, 32, 256, 128)):
    x = torch.randn(B, S, S, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
    with torch.no_grad():
        ref = ops.conv3x3_dense(x, w, None)
        row = []
        for cfg, sp in ((-1, 0), (0, 1), (0, 2), (0, 4), (1, 1), (1, 2), (2, 1), (2, 2), (2, 4)):
            call("mdvit_gemm_force_plan", cfg, sp)
            try:
                y = ops.conv3x3_dense(x, w, None)
                ok = float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
                t = timed(lambda: ops.conv3x3_dense(x, w, None))
                row.append(f"cfg {cfg} sp {sp}: {t:6.1f}{'' if ok else ' BAD'}")
            except Exception as e:
                row.append(f"cfg {cfg} sp {sp}: {type(e).__name__}")
            finally:
                call("mdvit_gemm_force_plan", -1, 0)
    fl = 2.0 * B * S * S * Cout * Cin * 9
    print(f"B={B} {S}x{S} {Cin}->{Cout} ({fl / 1e9:.1f} GF): " + " | ".join(row), flush=True)
