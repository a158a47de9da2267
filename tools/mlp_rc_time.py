"""Kernel times of the fused MLP entry points (csrc/mlp_rc.hip), called through the C ABI with HIP events around N back-to-back launches:
    C = 64 :  mdvit_mlp_rc_fwd / _dgrad / _wgrad          (hidden 512)      default 524288 tokens (stage 0 of the 32-image block)
    C = 128:  mdvit_mlp_rc16_fwd / _dgrad                  (hidden 1024)     default 131072 tokens (stage 1)
python tools/mlp_rc_time.py [--tokens64 N] [--tokens128 N] [--drop 0.1] [--save out.pt]
--save writes the outputs of a small fixed-seed problem (for a bit-for-bit comparison of two library builds: MDVIT_HIP_LIB=... python tools/mlp_rc_time.py --save a.pt)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import ops, _lib
from mdvit_amd._lib import call

ap = argparse.ArgumentParser()
ap.add_argument("--tokens64", type=int, default=524288)
ap.add_argument("--tokens128", type=int, default=131072)
ap.add_argument("--drop", type=float, default=0.1)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--save", default=None)
ap.add_argument("--planes", type=int, default=2, help="mdvit_mlp_rc_planes: 2 = bf16x3 (parity), 1 = one bf16 plane per operand (the bf16 speed mode)")
ap.add_argument("--variant", type=int, default=0, help="mdvit_mlp_rc_config value (C = 64 forward kernel variant)")
a = ap.parse_args()
_p, dev = ops._p, "cuda"


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


def problem(M, C, Hd, seed):
    g_ = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g_) * sc).to(dev)
    return dict(x=r(M, C), res=r(M, C), gm=r(M, C), W1=r(Hd, C, sc=C ** -0.5), b1=r(Hd, sc=0.1), W2=r(C, Hd, sc=Hd ** -0.5), b2=r(C, sc=0.1))


def entries(M, C, Hd, drop, t):
    st = ops._stream()
    W1p, W2p = ops._wplanes(t["W1"], False, 2), ops._wplanes(t["W2"], False, 2)
    W2tp, W1tp = ops._wplanes(t["W2"], True, 2), ops._wplanes(t["W1"], True, 2)
    y, dx = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
    k = (11, 22, 33, 44)
    out = {"y": y, "dx": dx}
    if C == 64:
        dW1, db1, dW2 = torch.empty(Hd, C, device=dev), torch.empty(Hd, device=dev), torch.empty(C, Hd, device=dev)
        wsb = _lib.load().mdvit_mlp_rc_wgrad_ws_bytes(M, C, Hd)
        ws = torch.empty(wsb // 4, device=dev)
        out.update(dW1=dW1, db1=db1, dW2=dW2)
        fns = {
            "fwd": lambda: call("mdvit_mlp_rc_fwd", _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2p), _p(t["b2"]), _p(t["res"]), None, M, _p(y), M, C, Hd, drop, *k, None, st),
            "dgrad": lambda: call("mdvit_mlp_rc_dgrad", _p(t["gm"]), _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2tp), _p(W1tp), _p(dx), M, C, Hd, drop, k[0], k[1], None, st),
            "wgrad+reduce": lambda: call("mdvit_mlp_rc_wgrad", _p(t["gm"]), _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2tp), _p(dW1), _p(db1), _p(dW2), _p(ws), wsb, M, C, Hd,
                                         drop, k[0], k[1], None, 0, st),
        }
        if hasattr(_lib.load(), "mdvit_mlp_rc_bwd"):
            parts = torch.empty(Hd // 256, M, C, device=dev)
            fns["bwd fused (dx parts + dW, +reduce)"] = lambda: call("mdvit_mlp_rc_bwd", _p(t["gm"]), _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2tp), _p(W1tp), _p(parts), _p(dW1),
                                                                      _p(db1), _p(dW2), _p(ws), wsb, M, C, Hd, drop, k[0], k[1], None, 0, st)
            fns["sum of the dx parts"] = lambda: call("mdvit_sum_batch", _p(parts), _p(dx), Hd // 256, M * C, st)
    else:
        h, du = torch.empty(M, Hd, device=dev), torch.empty(M, Hd, device=dev)
        out.update(h=h, du=du)
        fns = {
            "fwd16 (+h)": lambda: call("mdvit_mlp_rc16_fwd", _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2p), _p(t["b2"]), _p(t["res"]), None, M, _p(h), _p(y), M, C, Hd, drop, *k, None, st),
            "dgrad16 (+du)": lambda: call("mdvit_mlp_rc16_dgrad", _p(t["gm"]), _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2tp), _p(W1tp), _p(du), _p(dx), M, C, Hd, drop, k[0], k[1], None, st),
            "dgrad16 (no du)": lambda: call("mdvit_mlp_rc16_dgrad", _p(t["gm"]), _p(t["x"]), _p(W1p), _p(t["b1"]), _p(W2tp), _p(W1tp), None, _p(dx), M, C, Hd, drop, k[0], k[1], None, st),
        }
    return fns, out, (W1p, W2p, W2tp, W1tp)


print(f"library: {_lib.LIB_PATH}  variant {a.variant}  planes {a.planes}", flush=True)
if a.variant:
    call("mdvit_mlp_rc_config", a.variant)
call("mdvit_mlp_rc_planes", a.planes)
if a.save:
    saved = {}
    for C, Hd, M in ((64, 512, 4173), (128, 1024, 1031)):
        for drop in (0.0, a.drop):
            fns, out, keep = entries(M, C, Hd, drop, problem(M, C, Hd, 7))
            for f in fns.values():
                f()
            torch.cuda.synchronize()
            for n, v in out.items():
                saved[f"C{C}_p{drop}_{n}"] = v.cpu().clone()
    torch.save(saved, a.save)
    print(f"saved {len(saved)} tensors to {a.save}")
for C, Hd, M in ((64, 512, a.tokens64), (128, 1024, a.tokens128)):
    if M <= 0:
        continue
    t = problem(M, C, Hd, 3)
    for drop in (a.drop, 0.0):
        fns, out, keep = entries(M, C, Hd, drop, t)
        for rnd in range(a.rounds):
            line = "   ".join(f"{n} {timed(f):7.1f}" for n, f in fns.items())
            print(f"C={C} tokens={M} drop={drop} round {rnd}:  {line}  us", flush=True)
