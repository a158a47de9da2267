"""Roofline of ONE MHSA + Domain-Adapter block (SerialBlock_adapt, mdvit.py:316-361) forward + backward on the GPU.

BASELINE.json's target is quoted on "the fused MHSA+DA block at bs=32, 512x512": this tool times that block alone at
the four encoder stage shapes (tokens N = (512/4/2^s)^2, C = 64/128/320/512, 8 heads, MLP ratio 8/8/4/4) and prices it
against the chip roofs.  The bound is the sum over the block's operators of max(bytes / 8 TB/s, flops / MFMA roof),
bytes = the operator's compulsory fp32 inputs + outputs (the layout in HBM; nothing is counted twice inside an
operator, nothing is assumed fused across operators beyond what the reference math allows: bias / GELU / dropout /
DropPath / residual ride in the GEMM that produces the tensor, the Domain Adapter rides in the attention kernel).
Next to it: SURVEY.md 8(d)'s much stricter per-block figure -- the whole block as ONE fused bf16 kernel.

    python tools/block_roofline.py [--batch 32] [--iters 10] [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM = 8.0e12
MFMA = {"bf16x3": 2500e12 / 3.0, "fp32": 157.3e12, "bf16": 2500e12}       # useful flops/s (bf16x3 spends 3 bf16 MFMAs per product)
ATT_MFMA = 157.3e12                                      # the attention's small products run on fp32 MFMA


def block_ops(T, C, Hd, heads):
    """(name, floats moved, flops, mfma roof key) per operator of the block; T = B*N tokens."""
    Ch = C // heads
    taps = 60.0          # 2 * mean window taps per channel: heads 2x(3x3) + 3x(5x5) + 3x(7x7) -> 30 MACs
    f = []
    # ---- forward
    f += [("cpe dw3x3 + x", 2 * T * C, 18 * T * C, None)]
    f += [("LN1", 2 * T * C, 8 * T * C, None)]
    f += [("qkv GEMM", T * C + 3 * C * C + 3 * T * C, 6 * T * C * C, "gemm")]
    f += [("factor-att + crpe + DA", 4 * T * C, 4 * T * C * Ch + taps * T * C, "att")]
    f += [("proj GEMM + drop + res", 3 * T * C + C * C, 2 * T * C * C, "gemm")]
    f += [("LN2", 2 * T * C, 8 * T * C, None)]
    f += [("fc1 GEMM + GELU", T * C + C * Hd + T * Hd, 2 * T * C * Hd, "gemm")]
    f += [("fc2 GEMM + drop + res", T * Hd + C * Hd + 2 * T * C, 2 * T * C * Hd, "gemm")]
    b = []
    # ---- backward
    b += [("fc2 dgrad (x gelu')", T * C + C * Hd + 2 * T * Hd, 2 * T * C * Hd, "gemm")]
    b += [("fc2 wgrad", T * Hd + T * C + C * Hd, 2 * T * C * Hd, "gemm")]
    b += [("fc1 dgrad", T * Hd + C * Hd + T * C, 2 * T * C * Hd, "gemm")]
    b += [("fc1 wgrad", T * C + T * Hd + C * Hd, 2 * T * C * Hd, "gemm")]
    b += [("LN2 bwd (+res grad)", 4 * T * C, 12 * T * C, None)]
    b += [("proj dgrad", 2 * T * C + C * C, 2 * T * C * C, "gemm")]
    b += [("proj wgrad", 2 * T * C + C * C, 2 * T * C * C, "gemm")]
    b += [("factor-att + crpe + DA bwd", 7 * T * C, 8 * T * C * Ch + 2 * taps * T * C, "att")]
    b += [("qkv dgrad", 3 * T * C + 3 * C * C + T * C, 6 * T * C * C, "gemm")]
    b += [("qkv wgrad", T * C + 3 * T * C + 3 * C * C, 6 * T * C * C, "gemm")]
    b += [("LN1 bwd (+res grad)", 4 * T * C, 12 * T * C, None)]
    b += [("cpe bwd", 2 * T * C, 36 * T * C, None)]
    return f, b


def block_ops_strict(T, C, Hd, heads):
    """the same operator list with the MLP FUSED: nothing of size T x hidden moves -- forward x, res in / y out; backward data path gm, x in /
    dx out; weight gradients x, gm in (the operator-sum bound above charges the unfused fp32 T x hidden write and three
    re-reads as compulsory, 0.94 of the 1.83 ms stage-0 bound).  Flops stay the algorithmic ones: recomputation is not useful work."""
    f, b = block_ops(T, C, Hd, heads)
    f = [o for o in f if not o[0].startswith("fc")] + [("MLP fused (x, res -> y)", 3 * T * C + 2 * C * Hd, 4 * T * C * Hd, "gemm")]
    b = [o for o in b if not o[0].startswith("fc")] + [("MLP dgrad fused (gm, x -> dx)", 3 * T * C + 3 * C * Hd, 4 * T * C * Hd, "gemm"),
                                                       ("MLP wgrad fused (gm, x -> dW1, dW2)", 2 * T * C + 2 * C * Hd, 4 * T * C * Hd, "gemm")]
    return f, b


def survey_bound(B, N, C, r, heads, num_domains=4):
    """SURVEY.md 8(d) "per-block figures": the WHOLE block fused, bf16 storage, QKV materialised once ("two-phase").
    fwd: MACs = (4+2r) N C^2 + 2 N C Ch + 9 N C + 240 N Ch + (4 hid + hid C) per image, bytes = (8 N C B + params) * 2;
    bwd ~ 2x the flops, 3x the activation bytes.  Roofs: 2.5 PFLOP/s dense bf16, 8 TB/s."""
    Ch, hid = C // heads, max(C // 2, 4)
    macs = (4 + 2 * r) * N * C * C + 2 * N * C * Ch + 9 * N * C + 240 * N * Ch + (num_domains * hid + hid * C)
    params = (4 + 2 * r) * C * C + (5 + r) * C + 4 * C + num_domains * hid + hid + hid * C + C
    act = 8.0 * N * C * B * 2.0
    fwd = max(2.0 * macs * B / 2500e12, (act + 2.0 * params) / HBM)
    bwd = max(4.0 * macs * B / 2500e12, (3.0 * act + 2.0 * params) / HBM)
    return fwd, bwd


def bound_seconds(ops_list, precision):
    """precision "bf16": the bound of the mode BASELINE configs[1] / [3] name -- every tensor at 2 bytes per element, every product (the attention's small ones too) on the
    dense bf16 MFMA roof -- whatever the implementation stores today (it keeps fp32 activations except the C = 128 MLP's hidden tensors: the gap is the point of the figure)."""
    eb = 2.0 if precision == "bf16" else 4.0
    tot = 0.0
    rows = []
    for name, floats, flops, kind in ops_list:
        tb = eb * floats / HBM
        tf = flops / (MFMA[precision] if (kind == "gemm" or precision == "bf16") else ATT_MFMA) if kind else 0.0
        tot += max(tb, tf)
        rows.append({"op": name, "bytes": eb * floats, "flops": float(flops), "bound_us": 1e6 * max(tb, tf), "by": "hbm" if tb >= tf else "mfma"})
    return tot, rows


def stage0_mlp_kernels(B, img, drop, iters=10):
    """The largest kernels of the stage-0 block at this batch -- the fused-MLP kernels of mlp_rc.hip -- each ALONE on the GPU: the C entry points called back to back with
    events around `iters` launches (round 5; rounds 3-4 timed ops.mlp_residual's Python calls and took differences of three passes: the forward figure carried ~60 us of host
    work per call).  Priced against BOTH chip roofs with their ALGORITHMIC bytes / flops (recomputation is not useful work); what bounds them is neither: see `bound`."""
    from mdvit_amd import ops, _lib
    from mdvit_amd._lib import call
    dev = torch.device("cuda:0")
    side = img // 4
    T, C, Hd = B * side * side, 64, 512
    g_ = torch.Generator(device="cpu").manual_seed(3)
    r = lambda *s_, sc=1.0: (torch.randn(*s_, generator=g_) * sc).to(dev)
    x, res, gm = r(T, C), r(T, C), r(T, C)
    W1, b1, W2, b2 = r(Hd, C, sc=C ** -0.5), r(Hd, sc=0.1), r(C, Hd, sc=Hd ** -0.5), r(C, sc=0.1)
    rs = (torch.rand(B, generator=g_) < 0.9).float().to(dev) / 0.9
    W1p, W2p, W2tp, W1tp = ops._wplanes(W1, False, 2), ops._wplanes(W2, False, 2), ops._wplanes(W2, True, 2), ops._wplanes(W1, True, 2)
    y, dx, parts = torch.empty(T, C, device=dev), torch.empty(T, C, device=dev), torch.empty(Hd // 256, T, C, device=dev)
    dW1, db1, dW2 = torch.empty(Hd, C, device=dev), torch.empty(Hd, device=dev), torch.empty(C, Hd, device=dev)
    wsb = _lib.load().mdvit_mlp_rc_wgrad_ws_bytes(T, C, Hd)
    ws = torch.empty(wsb // 4, device=dev)
    P, st, k = ops._p, ops._stream(), (11, 22, 33, 44)

    def timed(fn):
        for _ in range(30):                  # (the first ~30 launches after an idle stretch run ~20 % slower: the clock is still ramping -- tools/mlp_rc_time.py's round 0)
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    t_f = timed(lambda: call("mdvit_mlp_rc_fwd", P(x), P(W1p), P(b1), P(W2p), P(b2), P(res), P(rs), side * side, P(y), T, C, Hd, drop, *k, None, st))
    t_d = timed(lambda: call("mdvit_mlp_rc_dgrad", P(gm), P(x), P(W1p), P(b1), P(W2tp), P(W1tp), P(dx), T, C, Hd, drop, k[0], k[1], None, st))
    t_w = timed(lambda: call("mdvit_mlp_rc_wgrad", P(gm), P(x), P(W1p), P(b1), P(W2tp), P(dW1), P(db1), P(dW2), P(ws), wsb, T, C, Hd, drop, k[0], k[1], None, 0, st))
    t_b = timed(lambda: call("mdvit_mlp_rc_bwd", P(gm), P(x), P(W1p), P(b1), P(W2tp), P(W1tp), P(parts), P(dW1), P(db1), P(dW2), P(ws), wsb, T, C, Hd, drop, k[0], k[1], None, 0, st))
    rows = []
    for name, secs, nbytes, flops in (("mlp_rc_fwd3_kernel (x, res -> y)", t_f, 4.0 * (3 * T * C + 2 * C * Hd), 4.0 * T * C * Hd),
                                      ("mlp_rc_dgrad_kernel (gm, x -> dx)", t_d, 4.0 * (3 * T * C + 3 * C * Hd), 4.0 * T * C * Hd),
                                      ("mlp_rc_wgrad_kernel + rc_reduce (gm, x -> dW1, db1, dW2)", t_w, 4.0 * (2 * T * C + 2 * C * Hd), 4.0 * T * C * Hd),
                                      ("mlp_rc_wgrad_kernel<DX> + rc_reduce = the whole backward (gm, x -> dx parts, dW1, db1, dW2)", t_b, 4.0 * (4 * T * C + 3 * C * Hd), 8.0 * T * C * Hd)):
        rows.append({"kernel": name, "us": secs * 1e6, "algorithmic_bytes": nbytes, "algorithmic_flop": flops, "hbm_GBps": nbytes / secs / 1e9,
                     "frac_hbm": nbytes / secs / HBM, "useful_TFLOPs": flops / secs / 1e12, "frac_mfma_bf16x3": flops / secs / MFMA["bf16x3"],
                     "bound": "VALU + MFMA in series (on one SIMD an MFMA does not run beside a saturated VALU pipe: profiles/r05_mfma_valu_overlap.txt; erf-GELU / dropout / "
                              "hi-lo split ~70 VALU cycles per hidden element + 48 MFMA cycles of bf16x3 products) -- neither chip roof; see DESIGN.md section 3"})
    return {"rows": T, "C": C, "hidden": Hd, "timer": "torch events around 10 back-to-back launches of the C entry point on the launch stream, nothing else on the GPU",
            "kernels": rows, "largest": max(rows[:3], key=lambda r_: r_["us"])}


def stage0_attention_core(B, img, iters=10):
    """The attention core of the stage-0 block (mdvit_factoratt_fwd / _bwd through the C ABI: factorised attention + ConvRelPosEnc + Domain Adapter scale, mdvit.py:281-313)
    alone on the GPU, against the bytes a two-pass fused form must move: forward q, k, v in / out written (16 T C bytes: SURVEY 8(d)'s 8 N C B at 2 bytes, here fp32);
    backward dout, q, k, v in / dq, dk, dv out (28 T C bytes).  What the kernels move on top of that (U written and re-read, dU, conv^T(dU), the tile partials) is the gap."""
    from mdvit_amd import _lib
    from mdvit_amd._lib import call
    C, heads, H = 64, 8, img // 4
    N, Ch, dev = H * H, C // heads, "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
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
    P = lambda t: None if t is None else t.data_ptr()

    def fwd():
        call("mdvit_factoratt_fwd", P(qkv), P(ws_[0]), P(bs_[0]), P(ws_[1]), P(bs_[1]), P(ws_[2]), P(bs_[2]), P(a), P(out), P(U), P(kmax), P(ksum), P(Mmat), P(ws), wsb,
             B, H, H, C, heads, s3, s5, s7, st)

    def bwd():      # the data path of the block entry (the window-weight gradients run on the side stream: mdvit_factoratt_wgrad)
        call("mdvit_factoratt_bwd", P(dout), P(qkv), P(out), P(U), P(ws_[0]), P(bs_[0]), P(ws_[1]), P(bs_[1]), P(ws_[2]), P(bs_[2]), P(a), P(kmax), P(ksum), P(Mmat),
             P(dqkv), P(e), None, None, None, None, None, None, P(ws), wsb, B, H, H, C, heads, s3, s5, s7, st)

    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            fn()
        t1.record(); torch.cuda.synchronize()
        res[name] = t0.elapsed_time(t1) / iters * 1e-3
    T = B * N
    rows = []
    for name, secs, nbytes in (("factor-att + crpe + DA forward (qkv -> out)", res["fwd"], 16.0 * T * C), ("factor-att + crpe + DA backward (dout, qkv -> dqkv)", res["bwd"], 28.0 * T * C)):
        rows.append({"kernel": name, "us": secs * 1e6, "algorithmic_bytes": nbytes, "hbm_GBps": nbytes / secs / 1e9, "frac_hbm": nbytes / secs / HBM, "bound": "hbm"})
    return {"rows": T, "C": C, "timer": "torch events around mdvit_factoratt_fwd / _bwd (C ABI) on the launch stream, nothing else on the GPU", "passes": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--drop", type=float, default=0.1, help="drop_rate = drop_path_rate of the block (the train scripts' 0.1)")
    ap.add_argument("--precision", choices=["bf16x3", "fp32", "bf16"], default="bf16x3")
    ap.add_argument("--json", default=None)
    ap.add_argument("--stages", default="0,1,2,3")
    ap.add_argument("--eager", action="store_true", help="time eager launches (host-bound on the small stages) instead of HIP-graph replays")
    args = ap.parse_args()

    from mdvit_amd import ops
    from mdvit_amd.blocks import MHSA_stage_adapt, init_weights_
    ops.set_gemm_precision(args.precision)
    dev = torch.device("cuda:0")
    dims, ratios, heads = [64, 128, 320, 512], [8, 8, 4, 4], 8
    B = args.batch
    out = {"batch": B, "img": args.img, "precision": args.precision, "hbm_peak": HBM, "mfma_peak": MFMA[args.precision], "stages": []}
    tot_fb = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    for s, (C, r) in enumerate(zip(dims, ratios)):
        if str(s) not in args.stages.split(","):
            continue
        side = args.img // 4 // (2 ** s)
        N, T, Hd = side * side, B * side * side, C * r
        torch.manual_seed(10 + s)
        st = MHSA_stage_adapt(N, C, 1, heads, r, qkv_bias=True, drop_rate=args.drop, attn_drop_rate=0.0, drop_path_rate=args.drop,
                              num_domains=4, adapt_method="Sup")
        init_weights_(st)
        st = st.to(dev).train()
        blk = st.mhca_blks[0]
        x = torch.randn(B, N, C, device=dev).requires_grad_(True)
        lab = F.one_hot(torch.full((B,), 1), 4).float().to(dev)
        g = torch.randn(B, N, C, device=dev)

        def fwd():
            return blk(x, (side, side), lab)

        wstream = torch.cuda.Stream()         # warm up on a side stream, as torch asks before a capture that runs backward
        wstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(wstream):
            for _ in range(args.warmup):
                y = fwd(); y.backward(g)
                ops.refresh_transposes()
        torch.cuda.current_stream().wait_stream(wstream)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        if args.eager:
            for _ in range(args.iters):
                for p in st.parameters():
                    p.grad = None
                x.grad = None
                ev[0].record(); y = fwd(); ev[1].record(); y.backward(g); ev[2].record()
                torch.cuda.synchronize()
                tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
            tf, tb = tf / args.iters * 1e-3, tb / args.iters * 1e-3
        else:
            # GPU time without the host's launch cadence: replay the forward, and forward + backward, as HIP graphs
            # (the training step keeps the queue full, so this -- not the eager time of an isolated block -- is what a
            # block costs inside it); backward = (forward + backward) - forward
            ops.enable_device_seed(True)
            times = []
            for with_bwd in (False, True):
                for p in st.parameters():
                    p.grad = None
                x.grad = None
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    if with_bwd:
                        y = fwd()
                        y.backward(g)
                    else:
                        with torch.no_grad():       # the same forward kernels; nothing is kept for a backward
                            y = fwd()
                gr.replay(); torch.cuda.synchronize()
                ev[0].record()
                for _ in range(args.iters):
                    gr.replay()
                ev[1].record()
                torch.cuda.synchronize()
                times.append(ev[0].elapsed_time(ev[1]) / args.iters * 1e-3)
                del gr
            tf, tb = times[0], times[1] - times[0]
        fo, bo = block_ops(T, C, Hd, heads)
        bf, rows_f = bound_seconds(fo, args.precision)
        bb, rows_b = bound_seconds(bo, args.precision)
        by = sum((2.0 if args.precision == "bf16" else 4.0) * o[1] for o in fo + bo)
        fl = sum(o[2] for o in fo + bo)
        sfo, sbo = block_ops_strict(T, C, Hd, heads)
        bfs, _ = bound_seconds(sfo, args.precision)
        bbs, _ = bound_seconds(sbo, args.precision)
        sf, sb = survey_bound(B, N, C, r, heads)
        rec = {"stage": s, "C": C, "tokens_per_image": N, "rows": T, "hidden": Hd,
               "survey_fused_bf16_bound_fwd_ms": sf * 1e3, "survey_fused_bf16_bound_bwd_ms": sb * 1e3,
               "frac_of_survey_fused_bf16_bound": (sf + sb) / (tf + tb),
               "fwd_ms": tf * 1e3, "bwd_ms": tb * 1e3, "bound_fwd_ms": bf * 1e3, "bound_bwd_ms": bb * 1e3,
               "frac_fwd": bf / tf, "frac_bwd": bb / tb, "frac": (bf + bb) / (tf + tb),
               "strict_bound_fwd_ms": bfs * 1e3, "strict_bound_bwd_ms": bbs * 1e3, "frac_strict": (bfs + bbs) / (tf + tb),
               "algorithmic_GB": by / 1e9, "GFLOP": fl / 1e9, "achieved_TBps": by / (tf + tb) / 1e12, "achieved_TFLOPs": fl / (tf + tb) / 1e12,
               "ops_fwd": rows_f, "ops_bwd": rows_b}
        out["stages"].append(rec)
        for i, v in enumerate((tf, tb, bf, bb, bfs, bbs)):
            tot_fb[i] += v
        print(f"stage {s}: C={C:4d} N={N:6d} rows={T:8d}  fwd {tf * 1e3:7.3f} ms (bound {bf * 1e3:6.3f}, {100 * bf / tf:4.1f}%)  "
              f"bwd {tb * 1e3:7.3f} ms (bound {bb * 1e3:6.3f}, {100 * bb / tb:4.1f}%)  block {100 * (bf + bb) / (tf + tb):4.1f}% of roofline, {100 * (bfs + bbs) / (tf + tb):4.1f}% of the STRICT (fused-MLP) bound  "
              f"[{by / (tf + tb) / 1e12:.2f} TB/s, {fl / (tf + tb) / 1e12:.1f} TF/s]  | SURVEY 8d fused-bf16 bound fwd {sf * 1e6:.0f} + bwd {sb * 1e6:.0f} us: "
              f"{100 * (sf + sb) / (tf + tb):.1f}%", flush=True)
        del st, blk, x, g, y
        torch.cuda.empty_cache()
    out["all_stages"] = {"fwd_ms": tot_fb[0] * 1e3, "bwd_ms": tot_fb[1] * 1e3, "bound_fwd_ms": tot_fb[2] * 1e3, "bound_bwd_ms": tot_fb[3] * 1e3,
                         "frac": (tot_fb[2] + tot_fb[3]) / (tot_fb[0] + tot_fb[1]), "frac_strict": (tot_fb[4] + tot_fb[5]) / (tot_fb[0] + tot_fb[1])}
    if args.precision == "bf16x3" and "0" in args.stages.split(","):
        try:
            out["stage0_mlp_kernels"] = stage0_mlp_kernels(B, args.img, args.drop)
            for r in out["stage0_mlp_kernels"]["kernels"]:
                print(f"   {r['kernel']:62s} {r['us']:7.1f} us  {r['hbm_GBps']:7.0f} GB/s ({100 * r['frac_hbm']:4.1f}% of HBM)  {r['useful_TFLOPs']:6.1f} TF/s useful ({100 * r['frac_mfma_bf16x3']:4.1f}% of the bf16x3 roof)")
        except Exception as exc:           # a report, never a reason to lose the block figures
            out["stage0_mlp_kernels"] = {"error": repr(exc)}
        try:
            out["stage0_attention_core"] = stage0_attention_core(B, args.img)
            for r in out["stage0_attention_core"]["passes"]:
                print(f"   {r['kernel']:62s} {r['us']:7.1f} us  {r['hbm_GBps']:7.0f} GB/s ({100 * r['frac_hbm']:4.1f}% of HBM)")
        except Exception as exc:
            out["stage0_attention_core"] = {"error": repr(exc)}
    print(f"all four stages: {100 * out['all_stages']['frac']:.1f}% of the operator-sum bound, {100 * out['all_stages']['frac_strict']:.1f}% of the strict (fused-MLP) bound")
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
