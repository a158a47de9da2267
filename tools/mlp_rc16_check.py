"""mdvit_mlp_rc16_dgrad (C = 128 MLP backward data path in one kernel) against the two data-gradient GEMMs it replaces: results and timing.
python tools/mlp_rc16_check.py [tokens ...]"""
import itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import ops


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
C, Hd = 128, 1024
W1 = (torch.randn(Hd, C, device="cuda") * C ** -0.5).requires_grad_(True); b1 = (torch.randn(Hd, device="cuda") * 0.1).requires_grad_(True)
W2 = (torch.randn(C, Hd, device="cuda") * Hd ** -0.5).requires_grad_(True); b2 = (torch.randn(C, device="cuda") * 0.1).requires_grad_(True)
sizes = [int(a) for a in sys.argv[1:]] or [4096 + 77, 65536, 131072]
for M in sizes:
    x = torch.randn(M, C, device="cuda", requires_grad=True); res = torch.randn(M, C, device="cuda"); g = torch.randn(M, C, device="cuda")
    rs = (torch.rand(4, device="cuda") < 0.9).float() / 0.9
    outs = {}
    for rc in (0, 1):
        ops._mlp_rc16 = bool(rc)
        ops._key_counter = itertools.count(5)

        def fwd():
            return ops.mlp_residual(x, res, W1, b1, W2, b2, rowscale=rs, drop_p=0.1, rows_per_scale=(M + 3) // 4)
        y = fwd()
        for t in (x, W1, b1, W2, b2):
            t.grad = None
        y.backward(g)
        outs[rc] = [y.detach().clone(), x.grad.clone(), W1.grad.clone(), b1.grad.clone(), W2.grad.clone(), b2.grad.clone()]
        with torch.no_grad():
            t_f = timed(fwd)

        def fb():
            yy = fwd(); yy.backward(g)
        t_fb = timed(fb, 5)
        ops.set_dgrad_only(True)
        t_fd = timed(fb, 5)
        ops.set_dgrad_only(False)
        print(f"M={M:7d} rc16={rc}: fwd {t_f:8.1f} us   fwd+bwd {t_fb:8.1f} us   fwd+dgrad-only bwd {t_fd:8.1f} us", flush=True)
    for name, a, b in zip(("y", "dx", "dW1", "db1", "dW2", "db2"), outs[0], outs[1]):
        err = float((a - b).abs().max() / b.abs().max())
        print(f"   {name}: rc16 vs GEMM path max rel diff {err:.2e}   finite {bool(torch.isfinite(b).all())}")
    # against fp64 (p = 0.1 masks are the same in both paths: compare the two errors)
ops._mlp_rc16 = True
# forward variants at C = 128 (32-token waves on 32x32x16 tiles against 16-token waves), interleaved in one process
from mdvit_amd._lib import call
M = sizes[-1]
x = torch.randn(M, C, device="cuda", requires_grad=True); res = torch.randn(M, C, device="cuda")
for rnd_ in range(3):
    for var in (16, 32):
        call("mdvit_mlp_rc_config", var)
        for dp in (0.1, 0.0):
            t = timed(lambda: ops.mlp_residual(x, res, W1, b1, W2, b2, rowscale=None, drop_p=dp, rows_per_scale=M))
            print(f"round {rnd_} fwd variant {var} drop {dp}: {t:8.1f} us (h stored)", flush=True)
call("mdvit_mlp_rc_config", 16)
