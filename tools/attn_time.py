"""Factorised attention + Domain Adapter core ALONE (mdvit_factoratt_fwd / _bwd through the C ABI, one stream, nothing else on the GPU) at the four
encoder-stage shapes of the 128-image step:  python tools/attn_time.py [--batch 32] [--stages 0,1,2,3] [--iters 20]
Prints the forward / backward time per call and the bytes a perfectly fused pass would move (fwd: qkv in, out + U written; bwd: dout, qkv, out, U in,
dqkv out) against them.  Per-kernel split: run it under `rocprofv3 --kernel-trace --stats` (tools/probe/attn_kernel_trace.sh)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdvit_amd import _lib  # noqa: E402
from mdvit_amd._lib import call  # noqa: E402

STAGES = [(64, 8, 128), (128, 8, 64), (320, 8, 32), (512, 8, 16)]      # C, heads, H = W at 512 x 512 (mdvit.py:700-720: dims / num_heads of MDViT-small)


def _p(t):
    return None if t is None else t.data_ptr()


def run(stage, B, iters, warmup):
    C, heads, H = STAGES[stage]
    N, Ch = H * H, C // heads
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(stage)
    qkv = torch.randn(B, N, 3 * C, device=dev, generator=g)
    s3, s5, s7 = 2, 3, 3
    ws_ = [torch.randn(n * Ch, 1, k, k, device=dev, generator=g) * 0.1 for n, k in ((s3, 3), (s5, 5), (s7, 7))]
    bs_ = [torch.randn(n * Ch, device=dev, generator=g) * 0.1 for n in (s3, s5, s7)]
    a = torch.softmax(torch.randn(B, heads, Ch, device=dev, generator=g), 1).reshape(B, C).contiguous()
    out = torch.empty(B, N, C, device=dev); U = torch.empty_like(out)
    kmax = torch.empty(B, C, device=dev); ksum = torch.empty_like(kmax); Mmat = torch.empty(B, C, Ch, device=dev)
    wsb = _lib.load().mdvit_factoratt_ws_bytes(B, N, C, heads)
    ws = torch.empty(wsb // 4, device=dev)
    dout = torch.randn(B, N, C, device=dev, generator=g)
    dqkv = torch.empty_like(qkv); e = torch.empty(B, C, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        call("mdvit_factoratt_fwd", _p(qkv), _p(ws_[0]), _p(bs_[0]), _p(ws_[1]), _p(bs_[1]), _p(ws_[2]), _p(bs_[2]), _p(a), _p(out), _p(U), _p(kmax), _p(ksum),
             _p(Mmat), _p(ws), wsb, B, H, H, C, heads, s3, s5, s7, st)

    def bwd():      # the data path of the block entry: window-weight gradients deferred (mdvit_factoratt_wgrad on the side stream)
        call("mdvit_factoratt_bwd", _p(dout), _p(qkv), _p(out), _p(U), _p(ws_[0]), _p(bs_[0]), _p(ws_[1]), _p(bs_[1]), _p(ws_[2]), _p(bs_[2]), _p(a), _p(kmax),
             _p(ksum), _p(Mmat), _p(dqkv), _p(e), None, None, None, None, None, None, _p(ws), wsb, B, H, H, C, heads, s3, s5, s7, st)

    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fn()
        t1.record()
        torch.cuda.synchronize()
        res[name] = t0.elapsed_time(t1) / iters * 1e3
    T = B * N
    fb = T * C * 4 * (3 + 2)            # qkv in; out, U out
    bb = T * C * 4 * (1 + 3 + 1 + 1 + 3)
    print(f"stage {stage}: C={C:4d} Ch={Ch:3d} tokens={T:7d}  fwd {res['fwd']:7.1f} us ({fb / res['fwd'] / 1e6:5.2f} TB/s of the fused pass's {fb / 1e6:6.0f} MB)   "
          f"bwd {res['bwd']:7.1f} us ({bb / res['bwd'] / 1e6:5.2f} TB/s of {bb / 1e6:6.0f} MB)", flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--stages", default="0,1,2,3")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--apply-tiles", type=int, default=0, help="32-token tiles per workgroup of the apply kernels (0: the launcher's rule)")
    ap.add_argument("--apply-mode", type=int, default=-1, help="mdvit_factoratt_config (load order of the backward's apply kernel, Ch = 8 / 16)")
    args = ap.parse_args()
    if args.apply_mode >= 0:
        call("mdvit_factoratt_config", args.apply_mode, args.apply_tiles)
    for s in (int(v) for v in args.stages.split(",")):
        run(s, args.batch, args.iters, args.warmup)
