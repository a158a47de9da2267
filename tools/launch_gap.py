"""GPU-side period of back-to-back dependent tiny kernels on one stream (queue pre-filled behind a spin kernel, so the host is
not the limit): what a launch costs the GPU even when it does nothing."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib, ops
lib = _lib.load()
x = torch.zeros(1024, device="cuda"); y = torch.zeros(1 << 22, device="cuda")
g = torch.ones(64, device="cuda"); b = torch.zeros(64, device="cuda")
xs = torch.randn(4096, 64, device="cuda"); ys = torch.empty_like(xs); mean = torch.empty(4096, device="cuda"); rstd = torch.empty(4096, device="cuda")
for name, fn, n in (("torch add (1024 floats)", lambda: x.add_(1.0), 2000),
                    ("layernorm 4096x64 through the C ABI", lambda: ops.call("mdvit_layernorm_fwd", ops._p(xs), ops._p(g), ops._p(b), ops._p(ys), ops._p(mean), ops._p(rstd), 4096, 64, 1, 1e-6, ops._stream()), 2000),
                    ("torch add (4M floats)", lambda: y.add_(1.0), 500)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(40_000_000)            # ~20 ms: the host enqueues everything behind it
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.2f} us per launch (GPU side)")
