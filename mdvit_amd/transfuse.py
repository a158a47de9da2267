"""TransFuse_S_adapt (BASELINE configs[4]) on the HIP kernels: the reference's module tree and state_dict (630 keys: `resnet.*`,
`transformer.*`, `up1` .. `up_c_2_2`, `final_*`) with forwards that run libmdvit_hip.so kernels on NHWC activations.

Reference: Models/Hybrid_models/TransFuseFolder/TransFuse.py:182-283 (model), :25-76 (BiFusion_block), :523-656 (Up, Attention_block,
DoubleConv, Residual, Conv); vision_transformer.py:125-214 (Attention_Sup, Block_adapt); DeiT.py:51-71,116-139; torchvision's
ResNet-34 (conv1 .. layer3); multi_train_TransFuse.py:29-38,141-189 (structure_loss, the step).
Dense 3x3 / 1x1 convolutions, BatchNorm(+ReLU), LayerNorm, Linear / MLP and the Domain Adapter reuse the MDViT ops; the rest is
csrc/transfuse.hip through the autograd Functions below.  Only 256x256 inputs are legal (pos_embed holds 16x16 tokens), as in the reference.
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Dict, Optional, Sequence

import torch
from torch import nn

from . import _lib, ops
from ._lib import ACT_NONE, ACT_RELU, call
from .blocks import BatchNormAct, ConvParams, LayerNormParams, LinearParams, _NoParams
from .ops import _c, _chk, _empty, _empty_like, _next_key, _p, _partials_ws, _seed_ptr, _stream


# ---------------------------------------------------------------------------------------------------------------------------------
# autograd Functions over csrc/transfuse.hip
# ---------------------------------------------------------------------------------------------------------------------------------
class _ImgConv(torch.autograd.Function):
    """ResNet conv1: 7x7 stride 2 pad 3 on the NCHW image -> NHWC (no gradient w.r.t. the image)"""

    @staticmethod
    def forward(ctx, img, w):
        ctx.set_materialize_grads(False)
        _chk(img, w)
        B, Cin, H, W_ = img.shape
        y = _empty((B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, w.shape[0]), device=img.device, dtype=torch.float32)
        call("mdvit_imgconv_fwd", _p(img), _p(w), _p(y), B, H, W_, Cin, w.shape[0], w.shape[-1], _stream())
        ctx.save_for_backward(img, w)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None or ops._dgrad_only:
            return None, None
        img, w = ctx.saved_tensors
        B, Cin, H, W_ = img.shape
        dw = _empty_like(w)
        wsp, wsb, _keep = _partials_ws(w[0].numel() * w.shape[0], g.device)
        call("mdvit_imgconv_wgrad", _p(img), _p(_c(g)), _p(dw), wsp, wsb, B, H, W_, Cin, w.shape[0], w.shape[-1], 0, _stream())
        return None, dw


def stem_conv7(img, w):
    """ResNet conv1 on the matrix cores: im2col of the NCHW image ([B Ho Wo, 148]: 147 columns + one of zeros) x the weight padded likewise;
    the padding / slicing of the weight is on the autograd tape (copies), so the weight gradient is the TN GEMM of ops.linear's backward."""
    B, Cin, H, W_ = img.shape
    Ho, Wo = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    K = Cin * w.shape[-1] * w.shape[-2]
    ldc = (K + 3) // 4 * 4
    col = torch.empty((B * Ho * Wo, ldc), device=img.device, dtype=torch.float32)
    call("mdvit_imgconv_im2col", _p(_c(img)), _p(col), B, H, W_, Cin, w.shape[-1], ldc, _stream())
    wp = torch.nn.functional.pad(w.reshape(w.shape[0], K), (0, ldc - K))
    return ops.linear(col, wp).view(B, Ho, Wo, w.shape[0])


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, H, W_, Cn = x.shape
        y = _empty((B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cn), device=x.device, dtype=torch.float32)
        idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
        call("mdvit_maxpool3x3s2_fwd", _p(x), _p(y), C.c_void_p(idx.data_ptr()), B, H, W_, Cn, _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (B, H, W_, Cn)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None
        (idx,) = ctx.saved_tensors
        B, H, W_, Cn = ctx.shape
        dx = _empty(ctx.shape, device=g.device, dtype=torch.float32)
        call("mdvit_maxpool3x3s2_bwd", _p(_c(g)), C.c_void_p(idx.data_ptr()), _p(dx), B, H, W_, Cn, _stream())
        return dx


class _ResizeAC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, H, W_, Cn = x.shape
        y = _empty((B, Ho, Wo, Cn), device=x.device, dtype=torch.float32)
        call("mdvit_resize_ac_fwd", _p(x), _p(y), B, H, W_, Ho, Wo, Cn, _stream())
        ctx.meta = (B, H, W_, Ho, Wo, Cn)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        B, H, W_, Ho, Wo, Cn = ctx.meta
        dx = _empty((B, H, W_, Cn), device=g.device, dtype=torch.float32)
        call("mdvit_resize_ac_bwd", _p(_c(g)), _p(dx), B, H, W_, Ho, Wo, Cn, _stream())
        return dx, None, None


def resize_ac(x, scale: int):
    return _ResizeAC.apply(_c(x), x.shape[1] * scale, x.shape[2] * scale)


class _AddRelu(torch.autograd.Function):
    """relu(a + b) (b optional)"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        _chk(a, b)
        y = _empty_like(a)
        call("mdvit_ew", _p(a), _p(b), _p(y), a.numel(), 0, _stream())
        ctx.save_for_backward(y)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        (y,) = ctx.saved_tensors
        dg = _empty_like(y)
        call("mdvit_ew", _p(_c(g)), _p(y), _p(dg), y.numel(), 2, _stream())
        return dg, (dg if ctx.has_b else None)


def add_relu(a, b=None):
    return _AddRelu.apply(_c(a), None if b is None else _c(b))


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        _chk(a, b)
        y = _empty_like(a)
        call("mdvit_ew", _p(a), _p(b), _p(y), a.numel(), 1, _stream())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        a, b = ctx.saved_tensors
        g = _c(g)
        da, db = _empty_like(a), _empty_like(b)
        call("mdvit_ew", _p(g), _p(b), _p(da), a.numel(), 1, _stream())
        call("mdvit_ew", _p(g), _p(a), _p(db), a.numel(), 1, _stream())
        return da, db


class _Gate(torch.autograd.Function):
    """y = sigmoid(s) * x; x [B, ..., C]; mode 0: s [B, P] per pixel, mode 1: s [B, C] per channel"""

    @staticmethod
    def forward(ctx, x, s, mode):
        ctx.set_materialize_grads(False)
        _chk(x, s)
        B, Cn = x.shape[0], x.shape[-1]
        P = x.numel() // (B * Cn)
        y = _empty_like(x)
        call("mdvit_gate_fwd", _p(x), _p(s), _p(y), B, P, Cn, mode, _stream())
        ctx.save_for_backward(x, s)
        ctx.meta = (B, P, Cn, mode)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        x, s = ctx.saved_tensors
        B, P, Cn, mode = ctx.meta
        dx, ds = _empty_like(x), _empty_like(s)
        wsb = _lib.load().mdvit_gate_bwd_ws_bytes(B, P, Cn, mode)
        ws = _empty((max(wsb // 4, 1),), device=x.device, dtype=torch.float32)
        call("mdvit_gate_bwd", _p(_c(g)), _p(x), _p(s), _p(dx), _p(ds), _p(ws), wsb, B, P, Cn, mode, _stream())
        return dx, ds, None


class _ChanPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        _chk(x)
        Cn = x.shape[-1]
        M = x.numel() // Cn
        y = _empty(tuple(x.shape[:-1]) + (2,), device=x.device, dtype=torch.float32)
        idx = torch.empty((M,), device=x.device, dtype=torch.int32)
        call("mdvit_chanpool_fwd", _p(x), _p(y), C.c_void_p(idx.data_ptr()), M, Cn, _stream())
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None
        (idx,) = ctx.saved_tensors
        dx = _empty(ctx.shape, device=g.device, dtype=torch.float32)
        call("mdvit_chanpool_bwd", _p(_c(g)), C.c_void_p(idx.data_ptr()), _p(dx), idx.numel(), ctx.shape[-1], _stream())
        return dx


class _Conv7x7_2to1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.set_materialize_grads(False)
        _chk(x, w)
        B, H, W_, _ = x.shape
        y = _empty((B, H, W_), device=x.device, dtype=torch.float32)
        call("mdvit_conv7x7_2to1_fwd", _p(x), _p(w), _p(y), B, H, W_, _stream())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        x, w = ctx.saved_tensors
        B, H, W_, _ = x.shape
        dx = _empty_like(x)
        dw = None if ops._dgrad_only else _empty_like(w)
        call("mdvit_conv7x7_2to1_bwd", _p(_c(g)), _p(x), _p(w), _p(dx), _p(dw), B, H, W_, _stream())
        return dx, dw


class _BN1(torch.autograd.Function):
    """BatchNorm2d(1) over a one-channel map"""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, training, eps, momentum, groups):
        ctx.set_materialize_grads(False)
        _chk(x, gamma, beta, rm, rv)
        y = _empty_like(x)
        stat = _empty((groups, 2), device=x.device, dtype=torch.float32)
        call("mdvit_bn1_fwd", _p(x), _p(gamma), _p(beta), _p(rm), _p(rv), C.c_void_p(nbt.data_ptr()) if nbt is not None else None, _p(y), _p(stat),
             x.numel(), groups, int(training), eps, momentum, _stream())
        ctx.save_for_backward(x, gamma, stat)
        ctx.training, ctx.groups = training, groups
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 10
        x, gamma, stat = ctx.saved_tensors
        dx = _empty_like(x)
        dgb = _empty((2,), device=x.device, dtype=torch.float32)
        call("mdvit_bn1_bwd", _p(_c(g)), _p(x), _p(gamma), _p(stat), _p(dx), _p(dgb), x.numel(), ctx.groups, int(ctx.training), _stream())
        if ops._dgrad_only:
            return (dx,) + (None,) * 9
        return (dx, dgb[0:1].clone(), dgb[1:2].clone()) + (None,) * 7


class _Subsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, H, W_, Cn = x.shape
        y = _empty((B, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cn), device=x.device, dtype=torch.float32)
        call("mdvit_subsample2", _p(x), _p(y), B, H, W_, Cn, 0, _stream())
        ctx.shape = (B, H, W_, Cn)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None
        B, H, W_, Cn = ctx.shape
        dx = _empty(ctx.shape, device=g.device, dtype=torch.float32)
        call("mdvit_subsample2", _p(_c(g)), _p(dx), B, H, W_, Cn, 1, _stream())
        return dx


class _AddPos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe):
        ctx.set_materialize_grads(False)
        _chk(x, pe)
        y = _empty_like(x)
        call("mdvit_add_bcast", _p(x), _p(pe), _p(y), x.shape[0], pe.numel(), _stream())
        ctx.pe_shape = tuple(pe.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        g = _c(g)
        dpe = None
        if not ops._dgrad_only and ctx.needs_input_grad[1]:
            dpe = _empty(ctx.pe_shape, device=g.device, dtype=torch.float32)
            call("mdvit_sum_batch", _p(g), _p(dpe), g.shape[0], dpe.numel(), _stream())
        return g, dpe


class _Dropout2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, key):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, Cn = x.shape[0], x.shape[-1]
        P = x.numel() // (B * Cn)
        y = _empty_like(x)
        call("mdvit_dropout2d", _p(x), _p(y), B, P, Cn, p, key[0], key[1], _seed_ptr(), _stream())
        ctx.meta = (B, P, Cn, p, key)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        B, P, Cn, p, key = ctx.meta
        dx = _empty_like(g)
        call("mdvit_dropout2d", _p(_c(g)), _p(dx), B, P, Cn, p, key[0], key[1], _seed_ptr(), _stream())
        return dx, None, None


def dropout2d(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    return _Dropout2d.apply(_c(x), float(p), _next_key())


_use_mfma_sdpa = os.environ.get("MDVIT_SDPA_MFMA", "1") != "0"         # A/B switch: 0 = the LDS-tiled fp32 VALU kernels at every N


class _SDPA(torch.autograd.Function):
    """Attention_Sup core + Domain Adapter (vision_transformer.py:148-169): qkv [B,N,3C] -> a * softmax(q k^T / sqrt(d)) v  [B,N,C]"""

    @staticmethod
    def forward(ctx, qkv, label, W1, b1, W2, b2, heads):
        ctx.set_materialize_grads(False)
        _chk(qkv, label, W1, b1, W2, b2)
        B, N, C3 = qkv.shape
        Cn = C3 // 3
        dev = qkv.device
        a = _empty((B, Cn), device=dev, dtype=torch.float32)
        call("mdvit_da_fwd", _p(label), _p(W1), _p(b1), _p(W2), _p(b2), _p(a), B, label.shape[1], W1.shape[0], Cn, heads, _stream())
        out = _empty((B, N, Cn), device=dev, dtype=torch.float32)
        ctx.mfma = N == 256 and heads <= 6 and Cn == heads * 64 and _use_mfma_sdpa          # the DeiT trunk's shape: fp32 matrix cores, no [N, N] tensor in HBM
        if ctx.mfma:
            P = _empty((B, heads, N), device=dev, dtype=torch.float32)      # row log-sum-exp
            call("mdvit_sdpa_mfma_fwd", _p(qkv), _p(a), _p(out), _p(P), B, N, Cn, heads, _stream())
        else:
            P = _empty((B, heads, N, N), device=dev, dtype=torch.float32)
            call("mdvit_sdpa_fwd", _p(qkv), _p(a), _p(out), _p(P), B, N, Cn, heads, _stream())
        ctx.save_for_backward(qkv, label, W1, b1, W2, b2, a, out, P)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 7
        qkv, label, W1, b1, W2, b2, a, out, P = ctx.saved_tensors
        heads = ctx.heads
        B, N, C3 = qkv.shape
        Cn = C3 // 3
        dev = qkv.device
        dqkv = _empty_like(qkv)
        e = _empty((B, Cn), device=dev, dtype=torch.float32)
        # mfma: scratch [2][B, heads, N] (row sums delta + partial column sums); else the [B, heads, N, N] score-gradient scratch
        dS = _empty((2,) + tuple(P.shape), device=dev, dtype=torch.float32) if ctx.mfma else _empty_like(P)
        call("mdvit_sdpa_mfma_bwd" if ctx.mfma else "mdvit_sdpa_bwd", _p(_c(g)), _p(qkv), _p(P), _p(out), _p(a), _p(dqkv), _p(e), _p(dS),
             B, N, Cn, heads, _stream())
        if ops._dgrad_only:
            return dqkv, None, None, None, None, None, None
        hid = W1.shape[0]
        dW1, db1, dW2, db2 = _empty_like(W1), _empty_like(b1), _empty_like(W2), _empty_like(b2)
        dab = _lib.load().mdvit_da_ws_bytes(B, hid, Cn)
        daws = _empty((dab // 4,), device=dev, dtype=torch.float32)
        call("mdvit_da_bwd", _p(label), _p(W1), _p(b1), _p(W2), _p(b2), _p(a), _p(e), 1.0, _p(dW1), _p(db1), _p(dW2), _p(db2), _p(daws), dab,
             B, label.shape[1], hid, Cn, heads, _stream())
        return dqkv, None, dW1, db1, dW2, db2, None


class _StructureLoss(torch.autograd.Function):
    """multi_train_TransFuse.py:29-38 on logits [B,1,H,W] (= NHWC with one channel), mask and the precomputed edge weights"""

    @staticmethod
    def forward(ctx, pred, mask, weit):
        ctx.set_materialize_grads(False)
        _chk(pred, mask, weit)
        B = pred.shape[0]
        HW = pred.numel() // B
        sums = _empty((B, 4), device=pred.device, dtype=torch.float64)
        loss = _empty((1,), device=pred.device, dtype=torch.float32)
        call("mdvit_structure_loss_fwd", _p(pred), _p(mask), _p(weit), C.c_void_p(sums.data_ptr()), _p(loss), B, HW, _stream())
        ctx.save_for_backward(pred, mask, weit, sums)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        pred, mask, weit, sums = ctx.saved_tensors
        B = pred.shape[0]
        dpred = _empty_like(pred)
        gs = g.reshape(1).float().contiguous()
        call("mdvit_structure_loss_bwd", _p(pred), _p(mask), _p(weit), C.c_void_p(sums.data_ptr()), _p(gs), _p(dpred), B, pred.numel() // B, _stream())
        return dpred, None, None


def structure_weight(mask):
    """weit = 1 + 5 |avg_pool2d(mask, 31, 1, 15) - mask| for a [B,1,H,W] mask (no gradient)"""
    mask = _c(mask.float())
    B, _, H, W_ = mask.shape
    tmp, weit = torch.empty_like(mask), torch.empty_like(mask)
    call("mdvit_structure_weight", _p(mask), _p(tmp), _p(weit), B, H, W_, _stream())
    return weit


def structure_loss(pred, mask, weit=None):
    mask = _c(mask.float())
    if weit is None:
        weit = structure_weight(mask)
    return _StructureLoss.apply(_c(pred), mask, weit)


# ---------------------------------------------------------------------------------------------------------------------------------
# modules (reference names; NHWC inside)
# ---------------------------------------------------------------------------------------------------------------------------------
def _conv(x, cp: ConvParams, stride: int = 1):
    k = cp.kernel_size[0]
    if k == 1:
        if stride == 2:
            x = _Subsample2.apply(_c(x))
        return ops.linear(x, cp.weight.view(cp.weight.shape[0], -1), cp.bias)
    assert k == 3
    return ops.conv3x3_dense(x, cp.weight, cp.bias, stride)


class BatchNorm1ch(nn.Module):
    """nn.BatchNorm2d(1) state; forward on a [B,H,W] map"""

    def __init__(self, eps=1e-5, momentum=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1)); self.bias = nn.Parameter(torch.zeros(1))
        self.register_buffer("running_mean", torch.zeros(1)); self.register_buffer("running_var", torch.ones(1))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = eps, momentum

    def forward(self, x):
        groups = ops._bn_groups if self.training else 1             # domain-batched forward: statistics per domain batch (ops.bn_groups)
        if x.shape[0] % groups:
            raise ValueError(f"bn_groups({groups}) needs a batch that is a multiple of it, got {x.shape[0]}")
        return _BN1.apply(_c(x), self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked, self.training, self.eps, self.momentum, groups)


class Conv(nn.Module):
    """TransFuse.py:639-656: conv (k = 1 / 3, 'same' padding) [+ BN] [+ ReLU]"""

    def __init__(self, inp_dim, out_dim, kernel_size=3, stride=1, bn=False, relu=True, bias=True):
        super().__init__()
        assert stride == 1 and kernel_size in (1, 3)
        self.conv = ConvParams(out_dim, inp_dim, kernel_size, kernel_size, bias=bias)
        self.bn = BatchNormAct(out_dim, ACT_RELU if relu else ACT_NONE) if bn else None
        self.relu = relu
        self.out_dim = out_dim

    def forward(self, x):
        if self.out_dim == 1 and self.conv.kernel_size[0] == 3:          # 3x3 conv to ONE channel: im2col + row dot
            B, H, W_, _ = x.shape
            col = ops._Im2col.apply(_c(x), 1, 1)
            y = ops.rowdot(col, self.conv.weight, self.conv.bias).view(B, H, W_, 1)
        else:
            y = _conv(x, self.conv)
        if self.bn is not None:
            return self.bn(y)
        return add_relu(y) if self.relu else y


class DoubleConv(nn.Module):                                       # TransFuse.py:579-598
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.double_conv = nn.Sequential(ConvParams(out_channels, in_channels, 3, 3), BatchNormAct(out_channels, ACT_RELU), _NoParams(),
                                         ConvParams(out_channels, out_channels, 3, 3), BatchNormAct(out_channels, ACT_NONE))
        self.identity = nn.Sequential(ConvParams(out_channels, in_channels, 1, 1), BatchNormAct(out_channels, ACT_NONE))

    def forward(self, x):
        dc, idt = self.double_conv, self.identity
        a = dc[1](_conv(x, dc[0]))
        a = dc[4](_conv(a, dc[3]))
        b = idt[1](_conv(x, idt[0]))
        return add_relu(a, b)


class Attention_block(nn.Module):                                  # TransFuse.py:552-576
    def __init__(self, F_g, F_l, F_int):
        super().__init__()
        self.W_g = nn.Sequential(ConvParams(F_int, F_g, 1, 1), BatchNormAct(F_int, ACT_NONE))
        self.W_x = nn.Sequential(ConvParams(F_int, F_l, 1, 1), BatchNormAct(F_int, ACT_NONE))
        self.psi = nn.Sequential(ConvParams(1, F_int, 1, 1), BatchNorm1ch(), _NoParams())

    def forward(self, g, x):
        g1 = self.W_g[1](_conv(g, self.W_g[0]))
        x1 = self.W_x[1](_conv(x, self.W_x[0]))
        psi = add_relu(g1, x1)
        s = self.psi[1](ops.rowdot(psi, self.psi[0].weight, self.psi[0].bias))          # [B,H,W] pre-sigmoid
        B = x.shape[0]
        return _Gate.apply(_c(x), s.reshape(B, -1), 0)                                    # x * sigmoid(psi)


class Up(nn.Module):                                               # TransFuse.py:523-549
    def __init__(self, in_ch1, out_ch, in_ch2=0, attn=False):
        super().__init__()
        self.conv = DoubleConv(in_ch1 + in_ch2, out_ch)
        self.attn_block = Attention_block(in_ch1, in_ch2, out_ch) if attn else None

    def forward(self, x1, x2=None):
        x1 = resize_ac(x1, 2)
        if x2 is not None:
            if x1.shape[1:3] != x2.shape[1:3]:
                raise ValueError("Up: sizes differ after the x2 upsample (the reference pads; at 256x256 they never differ)")
            if self.attn_block is not None:
                x2 = self.attn_block(x1, x2)
            x1 = torch.cat([x2, x1], dim=-1)
        return self.conv(x1)


class Residual(nn.Module):                                         # TransFuse.py:601-636
    def __init__(self, inp_dim, out_dim):
        super().__init__()
        half = out_dim // 2
        self.bn1 = BatchNormAct(inp_dim, ACT_RELU)
        self.conv1 = Conv(inp_dim, half, 1, relu=False)
        self.bn2 = BatchNormAct(half, ACT_RELU)
        self.conv2 = Conv(half, half, 3, relu=False)
        self.bn3 = BatchNormAct(half, ACT_RELU)
        self.conv3 = Conv(half, out_dim, 1, relu=False)
        self.skip_layer = Conv(inp_dim, out_dim, 1, relu=False)
        self.need_skip = inp_dim != out_dim

    def forward(self, x):
        residual = self.skip_layer(x) if self.need_skip else x
        out = self.conv1(self.bn1(x))
        out = self.conv2(self.bn2(out))
        out = self.bn3(out)
        c3 = self.conv3.conv                                        # conv3 (1x1) + residual in the GEMM epilogue
        return ops.linear(out, c3.weight.view(c3.weight.shape[0], -1), c3.bias, residual=residual)


class BiFusion_block(nn.Module):                                   # TransFuse.py:25-76
    def __init__(self, ch_1, ch_2, r_2, ch_int, ch_out, drop_rate=0.0):
        super().__init__()
        self.fc1 = ConvParams(ch_2 // r_2, ch_2, 1, 1)
        self.fc2 = ConvParams(ch_2, ch_2 // r_2, 1, 1)
        self.spatial = nn.Module()
        self.spatial.conv = ConvParams(1, 2, 7, 7, bias=False)
        self.spatial.bn = BatchNorm1ch()
        self.W_g = Conv(ch_1, ch_int, 1, bn=True, relu=False)
        self.W_x = Conv(ch_2, ch_int, 1, bn=True, relu=False)
        self.W = Conv(ch_int, ch_int, 3, bn=True, relu=True)
        self.residual = Residual(ch_1 + ch_2 + ch_int, ch_out)
        self.drop_rate = drop_rate

    def forward(self, g, x):
        B = g.shape[0]
        bp = self.W(_Mul.apply(_c(self.W_g(g)), _c(self.W_x(x))))
        # spatial attention for the CNN branch
        s = self.spatial.bn(_Conv7x7_2to1.apply(_ChanPool.apply(_c(g)), self.spatial.conv.weight))
        g = _Gate.apply(_c(g), s.reshape(B, -1), 0)
        # channel attention (SE) for the transformer branch
        t = ops.global_avg_pool(x)                                                       # [B, C]
        t = add_relu(ops.linear(t, self.fc1.weight.view(self.fc1.weight.shape[0], -1), self.fc1.bias))
        t = ops.linear(t, self.fc2.weight.view(self.fc2.weight.shape[0], -1), self.fc2.bias)
        x = _Gate.apply(_c(x), _c(t), 1)
        fuse = self.residual(torch.cat([g, x, bp], dim=-1))
        return dropout2d(fuse, self.drop_rate, self.training)


class BasicBlock(nn.Module):                                       # torchvision.models.resnet.BasicBlock
    def __init__(self, inp, out, stride=1):
        super().__init__()
        self.conv1 = ConvParams(out, inp, 3, 3, bias=False); self.bn1 = BatchNormAct(out, ACT_RELU)
        self.conv2 = ConvParams(out, out, 3, 3, bias=False); self.bn2 = BatchNormAct(out, ACT_NONE)
        self.downsample = nn.Sequential(ConvParams(out, inp, 1, 1, bias=False), BatchNormAct(out, ACT_NONE)) if (stride != 1 or inp != out) else None
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample[1](_conv(x, self.downsample[0], self.stride))
        y = self.bn1(_conv(x, self.conv1, self.stride))
        y = self.bn2(_conv(y, self.conv2))
        return add_relu(y, idt)


class ResNet34Trunk(nn.Module):
    """conv1 .. layer3 of torchvision's resnet34 (layer4 / fc are Identity in the reference, TransFuse.py:190-191)"""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvParams(64, 3, 7, 7, bias=False)
        self.bn1 = BatchNormAct(64, ACT_RELU)
        inp = 64
        for i, (c, n) in enumerate(((64, 3), (128, 4), (256, 6)), start=1):
            setattr(self, f"layer{i}", nn.Sequential(*([BasicBlock(inp, c, 1 if i == 1 else 2)] + [BasicBlock(c, c) for _ in range(n - 1)])))
            inp = c


class Attention_Sup(nn.Module):                                    # vision_transformer.py:125-169
    def __init__(self, dim, num_heads=8, r=2, num_domains=4):
        super().__init__()
        self.num_heads = num_heads
        hidden = max(dim // r, 4)
        self.qkv = LinearParams(dim, dim * 3, bias=True)
        self.proj = LinearParams(dim, dim)
        self.domain_layer = nn.Sequential(LinearParams(num_domains, hidden), _NoParams(), LinearParams(hidden, dim))

    def forward(self, x, domain_label, res):
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        d0, d2 = self.domain_layer[0], self.domain_layer[2]
        o = _SDPA.apply(_c(qkv), _c(domain_label.float()), d0.weight, d0.bias, d2.weight, d2.bias, self.num_heads)
        return ops.linear(o, self.proj.weight, self.proj.bias, residual=res)           # x + proj(.)  (drop rates are 0 in deit_small_adapt)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = LinearParams(dim, hidden); self.fc2 = LinearParams(hidden, dim)


class Block_adapt(nn.Module):                                      # vision_transformer.py:195-214
    def __init__(self, dim, num_heads, mlp_ratio=4.0, num_domains=4):
        super().__init__()
        self.norm1 = LayerNormParams(dim, 1e-6)
        self.attn = Attention_Sup(dim, num_heads, num_domains=num_domains)
        self.norm2 = LayerNormParams(dim, 1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, domain_label):
        # Round 6: the whole block as ONE C call per pass (csrc/block.hip, MdvitBlockDesc.attn_kind = 1: the kernels of the operator-level path below, in its order -- LN1,
        # qkv, adapter, softmax(q k^T) v on sdpa.hip, proj + residual, LN2, the two MLP products -- enqueued from C with the weight-gradient work forked onto the side
        # stream): 14 autograd nodes, ~25 library calls and as many torch.empty per block and step become 2 (the TransFuse step is bound by the host's enqueue time on the
        # slower hosts of the pool: profiles/r06_host_cprofile_transfuse.txt)
        a, m = self.attn, self.mlp
        d0, d2 = a.domain_layer[0], a.domain_layer[2]
        params = [None, None, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, None, None, None, None, None, None, d0.weight, d0.bias, d2.weight, d2.bias,
                  a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
        if domain_label is not None and ops.deit_block_entry_ok(x, a.num_heads, params) and ops.block_entry_ok(x.shape[-1], m.fc1.weight.shape[0], params):
            meta = (16, 16, a.num_heads, (0, 0, 0), float(self.norm1.eps), 0.0, 1, False, 1)
            return ops.serial_block(x, domain_label, None, None, meta, params)
        cur, x = self.norm1.fork(x)
        x = self.attn(cur, domain_label, x)
        cur, x = self.norm2.fork(x)
        return ops.mlp_residual(cur, x, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, None, 0.0, x.shape[1])


class _PatchEmbed(nn.Module):
    def __init__(self, embed_dim, patch=16):
        super().__init__()
        self.proj = ConvParams(embed_dim, 3, patch, patch)
        self.patch = patch


class DeiT_adapt(nn.Module):                                       # DeiT.py:51-71,116-139: deit_small_patch16_224_adapt, 16x16 tokens
    def __init__(self, embed_dim=384, depth=8, num_heads=6, mlp_ratio=4, num_domains=4):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))               # registered by the reference, unused by its forward
        self.pos_embed = nn.Parameter(torch.zeros(1, 256, embed_dim))
        self.patch_embed = _PatchEmbed(embed_dim)
        self.blocks = nn.ModuleList([Block_adapt(embed_dim, num_heads, mlp_ratio, num_domains) for _ in range(depth)])
        self.norm = LayerNormParams(embed_dim, 1e-6)

    def forward(self, img, domain_label):
        B, Cin, H, W_ = img.shape
        p = self.patch_embed.patch
        if (H // p) * (W_ // p) != self.pos_embed.shape[1]:
            raise ValueError(f"TransFuse_S_adapt accepts 256x256 inputs only (pos_embed holds {self.pos_embed.shape[1]} tokens; DeiT.py:134)")
        patches = torch.empty((B * (H // p) * (W_ // p), Cin * p * p), device=img.device, dtype=torch.float32)
        call("mdvit_patchify", _p(_c(img)), _p(patches), B, Cin, H, W_, p, _stream())
        w = self.patch_embed.proj
        x = ops.linear(patches, w.weight.view(w.weight.shape[0], -1), w.bias).view(B, -1, w.weight.shape[0])
        x = _AddPos.apply(_c(x), self.pos_embed)
        for blk in self.blocks:
            x = blk(x, domain_label)
        return self.norm(x)


class TransFuse_S_adapt(nn.Module):
    """TransFuse.py:182-283.  forward(imgs NCHW (B,3,256,256), domain_label (B,4)) -> (map_x, map_1, map_2), logits (B,1,256,256)."""

    def __init__(self, num_classes=1, drop_rate=0.2, normal_init=True, pretrained=False, pretrained_folder=None, num_domains=4):
        super().__init__()
        if num_classes != 1 or pretrained:
            raise NotImplementedError("num_classes = 1 without pretrained checkpoints is what the train script builds offline")
        self.resnet = ResNet34Trunk()
        self.transformer = DeiT_adapt(num_domains=num_domains)
        self.up1 = Up(384, 128); self.up2 = Up(128, 64)
        self.final_x = nn.Sequential(Conv(256, 64, 1, bn=True, relu=True), Conv(64, 64, 3, bn=True, relu=True), Conv(64, 1, 3, bn=False, relu=False))
        self.final_1 = nn.Sequential(Conv(64, 64, 3, bn=True, relu=True), Conv(64, 1, 3, bn=False, relu=False))
        self.final_2 = nn.Sequential(Conv(64, 64, 3, bn=True, relu=True), Conv(64, 1, 3, bn=False, relu=False))
        self.up_c = BiFusion_block(256, 384, 4, 256, 256, drop_rate / 2)
        self.up_c_1_1 = BiFusion_block(128, 128, 2, 128, 128, drop_rate / 2)
        self.up_c_1_2 = Up(256, 128, 128, attn=True)
        self.up_c_2_1 = BiFusion_block(64, 64, 1, 64, 64, drop_rate / 2)
        self.up_c_2_2 = Up(128, 64, 64, attn=True)
        self.drop_rate = drop_rate
        if normal_init:
            self.init_weights()

    def init_weights(self):
        """TransFuse.py:272-283,502-520: kaiming-normal (fan_in, relu) convolutions with zero bias, BatchNorm 1 / 0, on the decoder-side modules"""
        for top in (self.up1, self.up2, self.final_x, self.final_1, self.final_2, self.up_c, self.up_c_1_1, self.up_c_1_2, self.up_c_2_1, self.up_c_2_2):
            for m in top.modules():
                if isinstance(m, ConvParams):
                    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in", nonlinearity="relu")
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
                elif isinstance(m, (BatchNormAct, BatchNorm1ch)):
                    nn.init.ones_(m.weight); nn.init.zeros_(m.bias)

    def forward(self, imgs, domain_label, labels=None):
        B = imgs.shape[0]
        drop = lambda t: dropout2d(t, self.drop_rate, self.training)
        def transformer_branch():
            x_b = self.transformer(imgs, domain_label).view(B, 16, 16, -1)        # tokens ARE the NHWC map (the reference transposes + views)
            x_b = drop(x_b)
            x_b_1 = drop(self.up1(x_b))
            return x_b, x_b_1, drop(self.up2(x_b_1))
        # The DeiT branch (8 blocks over 256 tokens per image: chains of small kernels) and the ResNet branch (implicit-GEMM convolutions) share nothing
        # until the first BiFusion block: the DeiT branch runs on a stream of its own -- forward here, and its backward too (autograd runs a node's
        # backward on the stream of its forward and orders the streams at the graph's edges)
        def resnet_branch():
            r = self.resnet
            x_u = r.bn1(stem_conv7(imgs, r.conv1.weight))
            x_u = _MaxPool.apply(_c(x_u))
            x_u_2 = drop(r.layer1(x_u))
            x_u_1 = drop(r.layer2(x_u_2))
            return drop(r.layer3(x_u_1)), x_u_1, x_u_2
        bs = ops.branch_stream() if imgs.is_cuda else None
        if bs is None:
            x_b, x_b_1, x_b_2 = transformer_branch()
            x_u, x_u_1, x_u_2 = resnet_branch()
        else:
            main = torch.cuda.current_stream()
            bs.wait_stream(main)
            for t in (imgs, domain_label):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(bs)
            with torch.cuda.stream(bs):                   # (enqueued first; the ResNet branch first was measured slower: 1174 against 1207 images/s)
                x_b, x_b_1, x_b_2 = transformer_branch()
            for t in (x_b, x_b_1, x_b_2):
                t.record_stream(main)
            x_u, x_u_1, x_u_2 = resnet_branch()
            main.wait_stream(bs)
        x_c = self.up_c(x_u, x_b)
        x_c_1_1 = self.up_c_1_1(x_u_1, x_b_1)
        x_c_1 = self.up_c_1_2(x_c, x_c_1_1)
        x_c_2_1 = self.up_c_2_1(x_u_2, x_b_2)
        x_c_2 = self.up_c_2_2(x_c_1, x_c_2_1)

        def head(seq, x, scale):
            y = resize_ac(seq(x), scale)                                             # [B, 256, 256, 1] == NCHW (B,1,256,256)
            return y.view(B, 1, y.shape[1], y.shape[2])
        return head(self.final_x, x_c, 16), head(self.final_1, x_b_2, 4), head(self.final_2, x_c_2, 4)


def transfuse_train_step(model, batches: Sequence[tuple], optimizer=None, accumulator=None, num_domains: int = 4,
                         fuse_domains: bool = False) -> Dict[str, torch.Tensor]:
    """multi_train_TransFuse.py:141-189: per domain loss = 0.5 SL(map_2) + 0.3 SL(map_1) + 0.2 SL(map_x); ONE backward of the sum.
    batches: [(img (B,3,256,256), label (B,1,256,256), set_id (B,) int64 on the host)].
    fuse_domains: ONE forward over the concatenated, equally sized domain batches.  The reference runs one forward per domain; the only
    ops that couple the samples of a forward are the BatchNorms, and with ops.bn_groups(G) they keep statistics per domain batch (and
    update the running statistics G times in order), so the result is the same function with 1/G of the kernel launches (the step is
    launch-bound: ~5500 launches at 4 x 8 images)."""
    import torch.nn.functional as F
    ops.refresh_transposes()
    if accumulator is not None:
        accumulator.zero()
    elif optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    else:
        model.zero_grad(set_to_none=True)
    tot, per = None, []
    G = len(batches)
    if fuse_domains and G > 1 and all(b[0].shape == batches[0][0].shape for b in batches):
        img = torch.cat([b[0] for b in batches], 0)
        if all(len(b) > 3 and b[3] is not None for b in batches):       # one-hot labels already on the device (graph capture: no H2D copy)
            dl = torch.cat([b[3] for b in batches], 0)
        else:
            sid = torch.cat([b[2].cpu() for b in batches], 0)
            dl = F.one_hot(sid, num_domains).float().to(img.device, non_blocking=True)
        with ops.bn_groups(G):
            m4, m3, m2 = model(img, dl)
        parts = [ops.split_groups(t, G) for t in (m4, m3, m2)]
        for g_, (_, label, *_rest) in enumerate(batches):
            weit = structure_weight(label)
            loss = 0.5 * structure_loss(parts[2][g_], label, weit) + 0.3 * structure_loss(parts[1][g_], label, weit) + 0.2 * structure_loss(parts[0][g_], label, weit)
            per.append(loss.detach())
            tot = loss if tot is None else tot + loss
        batches = []
    for img, label, set_id, *pre in batches:
        dl = pre[0] if pre and pre[0] is not None else F.one_hot(set_id.cpu(), num_domains).float().to(img.device, non_blocking=True)
        m4, m3, m2 = model(img, dl)
        weit = structure_weight(label)
        loss = 0.5 * structure_loss(m2, label, weit) + 0.3 * structure_loss(m3, label, weit) + 0.2 * structure_loss(m4, label, weit)
        per.append(loss.detach())
        tot = loss if tot is None else tot + loss
    if accumulator is not None:
        accumulator.begin_sweep(True)
    ops.backward(tot)
    ops.join_side_stream()
    if accumulator is not None:
        accumulator.end_sweep(True)
    if optimizer is not None:
        optimizer.step()
    return {"loss": tot.detach(), "per_domain": torch.stack(per)}
