"""Decoder-side modules: U-Net decoding block with a transformer stage, the auxiliary "peer" head.

  UnetDecodingBlockTransformer  <- Models/Decoders.py:174-214
  MLPDecoderFM                  <- Models/Decoders.py:289-339

Two exact algebraic restructurings (both linear maps commute; bilinear weights sum to 1, so biases
commute too) keep the 512-channel full-resolution tensors of the reference out of HBM:
  * conv1x1(upsample(x)) == upsample(conv1x1(x))           (conv_before, finalconv, linear_out)
  * linear_fuse(cat_q upsample(linear_q(x_q))) == sum_q upsample((Wf_q W_q) x_q + Wf_q b_q)
The composed weights Wf_q W_q are formed by a GEMM on the autograd tape each call, so every
parameter still receives its exact gradient.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import ops
from ._lib import ACT_RELU
from .blocks import BatchNormAct, ConvParams, DecoderDWConv2d_BN, _NoParams, _check_norm


class UnetDecodingBlockTransformer(nn.Module):
    def __init__(self, in_channel, out_channel, mhsa_block, use_res=False, conv_norm=nn.BatchNorm2d, dsn=0):
        super().__init__()
        if use_res:
            raise NotImplementedError("use_res=True is never used by the reference's models")
        self.use_res = use_res
        self.conv_before = ConvParams(out_channel, in_channel, 1, 1, bias=True)
        self.conv_after = DecoderDWConv2d_BN(out_channel * 2, out_channel, norm_layer=conv_norm, dsn=dsn)
        self.mhsa_block = mhsa_block

    def forward(self, input, skip, domain_label=None):
        """input NHWC [B,h,w,Cin], skip NHWC [B,H,W,Cout] -> NHWC [B,H,W,Cout]."""
        B, H, W, Cout = skip.shape
        low = ops.linear(input, self.conv_before.weight, self.conv_before.bias)       # 1x1 conv at the LOW resolution
        up = ops.upsample_bilinear(low, H, W)
        z = self.conv_after(skip, up)
        t = self.mhsa_block(z.view(B, H * W, Cout), H, W, domain_label)
        return t.view(B, H, W, Cout)


_COMPOSE_GROUPED = os.environ.get("MDVIT_COMPOSE_GROUPED", "1") != "0"
_LOWS_GROUPED = os.environ.get("MDVIT_LOWS_GROUPED", "1") != "0"      # A/B: 0 = one launch per head for the low-resolution projections
_WT_BATCH = os.environ.get("MDVIT_WT_BATCH", "1") != "0"      # A/B: 0 = one transpose per use of a composed weight
# Round 6: the two full-resolution features of an 'MLPFM' head (the first encoder stage's and the main decoder's last: both H/4, 64 channels) go through ONE product
# over the concatenated K = C_0 + C_5 axis -- the [tokens, hidden] sum is written once in the forward (it was: written, read back as the second product's residual and
# written again) and its gradient is read by ONE data-gradient and ONE weight-gradient product per sweep instead of two each.  0: the two products (A/B)
_HEAD_CAT = os.environ.get("MDVIT_HEAD_CAT", "1") != "0"


class MLPDecoderFM(nn.Module):
    def __init__(self, in_channels, out_channel, hidden_channel=256, outfeature_channel=64, dropout_ratio=0.1, conv_norm=nn.BatchNorm2d):
        super().__init__()
        _check_norm(conv_norm)
        assert out_channel == 1
        self.linear1 = ConvParams(hidden_channel, in_channels[0], 1, 1)
        self.linear2 = ConvParams(hidden_channel, in_channels[1], 1, 1)
        self.linear3 = ConvParams(hidden_channel, in_channels[2], 1, 1)
        self.linear4 = ConvParams(hidden_channel, in_channels[3], 1, 1)
        self.linear_fuse = nn.Sequential(ConvParams(hidden_channel, hidden_channel * 4 + outfeature_channel, 1, 1),
                                         BatchNormAct(hidden_channel, ACT_RELU), _NoParams())
        self.dropout = nn.Dropout2d(dropout_ratio)      # holds p only; the mask is applied inside the BN+ReLU kernel
        self.linear_out = ConvParams(out_channel, hidden_channel, 1, 1)
        self.hidden = hidden_channel
        self.with_fm = outfeature_channel > 0

    def fuse_blocks(self):
        """the fuse weight as (its leading [hid, 4*hid] columns, the main-decoder feature's block or None)"""
        hid = self.hidden
        Wf = self.linear_fuse[0].weight.view(hid, -1)          # [hid, 4*hid + C5]
        if not self.with_fm:
            return Wf, None
        return ops.split_cols(Wf, [4 * hid, Wf.shape[1] - 4 * hid])

    @staticmethod
    def compose_many(heads):
        """the composed weights of SEVERAL heads in grouped launches (ops.compose_heads): per head (W5 block, [(Wc_q, bc_q)] * 4)"""
        blocks = [h.fuse_blocks() for h in heads]
        lins = [(h.linear1, h.linear2, h.linear3, h.linear4) for h in heads]
        if not _COMPOSE_GROUPED:        # A/B: one product + one row-dot launch per head and scale (rounds 2-3)
            hid = heads[0].hidden
            return [(blocks[g][1], [(ops.matmul(blocks[g][0][:, q * hid:(q + 1) * hid], l.weight.view(hid, -1)), ops.rowdot(blocks[g][0][:, q * hid:(q + 1) * hid], l.bias))
                                    for q, l in enumerate(lins[g])]) for g in range(len(heads))]
        comp = ops.compose_heads([b[0] for b in blocks], [[l.weight for l in ls] for ls in lins], [[l.bias for l in ls] for ls in lins])
        cat = _HEAD_CAT and all(h.with_fm for h in heads)
        if cat:
            # comp[g][4] = ([Wc_0 | W_5] [hid, C_0 + C_5], bc_0 + b_fuse): the operands of the one product over both full-resolution features (forward uses it
            # when features[0] and features[4] have one resolution)
            for g, h in enumerate(heads):
                comp[g].append((torch.cat([comp[g][0][0], blocks[g][1]], 1), comp[g][0][1] + h.linear_fuse[0].bias))
        if _WT_BATCH and ops._gemm_precision >= 1 and torch.is_grad_enabled():
            # the W^T every data-gradient product of the heads' 1x1 convolutions reads (both sweeps): one launch for all heads instead of one per use
            # (with the concatenated product: its [hid, C_0 + C_5] weight instead of the two blocks it replaces)
            ops.transpose_weights_batch([w for g in range(len(heads)) for (w, _) in (comp[g][1:] if cat else comp[g])] + ([] if cat else [blocks[g][1] for g in range(len(heads))]))
        return [(blocks[g][1], comp[g]) for g in range(len(heads))]

    @staticmethod
    def grouped_lows(composed, feats_per_head):
        """the low-resolution projections that run on the general GEMM (C_q not in {64, 128}: the streaming short-K Linear keeps the others) for ALL heads in one
        launch per scale (ops.linear_grouped); returns composed with a third entry per head: {q: projected feature}"""
        G = len(composed)
        if G < 2 or not _LOWS_GROUPED or any(c is None for c in composed):
            return composed
        h, w = feats_per_head[0][0].shape[1:3]
        out = [dict() for _ in range(G)]
        for q in range(4):
            f0 = feats_per_head[0][q]
            if (f0.shape[1] == h and f0.shape[2] == w) or f0.shape[-1] in (64, 128) or not f0.is_cuda:
                continue
            ys = ops.linear_grouped([feats_per_head[g][q] for g in range(G)], [composed[g][1][q][0] for g in range(G)], [composed[g][1][q][1] for g in range(G)])
            for g in range(G):
                out[g][q] = ys[g]
        return [(composed[g][0], composed[g][1], out[g]) for g in range(G)]

    def forward(self, features, img_size, out_feat=False, composed=None):
        if out_feat:
            raise NotImplementedError("out_feat=True of the aux head is unused by the train path")
        x1 = features[0]
        B, h, w, _ = x1.shape
        hid = self.hidden
        bias = self.linear_fuse[0].bias
        # fused = Wf_5 x5 + bf + sum_q upsample((Wf_q W_q) x_q + Wf_q b_q)      (MLPDecoder: no x5 term, bf rides on q = 0)
        W5, comp = (composed if composed is not None else MLPDecoderFM.compose_many([self])[0])[:2]
        pre = composed[2] if (composed is not None and len(composed) > 2) else {}       # q -> this head's projected feature, computed with the other heads' (grouped_lows)
        lows = []                                                                              # the projected lower-resolution features
        q0 = 0
        if self.with_fm and len(comp) > 4 and features[0].shape[1:3] == features[4].shape[1:3]:
            Wcat, bcat = comp[4]
            acc = ops.linear(ops.cat_channels(features[0], features[4]), Wcat, bcat)           # both full-resolution terms + both biases in ONE product: [B,h,w,hid]
            q0 = 1
        else:
            acc = ops.linear(features[4], W5, bias) if self.with_fm else None                  # [B,h,w,hid]
        for q in range(q0, 4):
            Wc, bc = comp[q]                                                                   # [hid, C_q], [hid]
            fq = features[q]
            if acc is None:
                assert fq.shape[1] == h and fq.shape[2] == w
                acc = ops.linear(fq, Wc, bc + bias)
            elif fq.shape[1] == h and fq.shape[2] == w:
                acc = ops.linear(fq, Wc, bc, residual=acc)
            elif q in pre:
                lows.append(pre[q])
            else:
                lows.append(ops.linear(fq, Wc, bc))
        if lows:
            acc = ops.upsample_sum(acc, lows, h, w)            # acc + sum_q resize(P_q): ONE pass over the [B,h,w,512] sum instead of one per source
        p = self.dropout.p if self.training else 0.0
        bn = self.linear_fuse[1]
        # BN + ReLU + Dropout2d + linear_out (512 -> 1) as ONE op: the normalised [B,h,w,512] tensor is never written
        low = ops.bn_act_rowdot(acc, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training, bn.act,
                                self.linear_out.weight, self.linear_out.bias, bn.eps, bn.momentum, p)          # [B,h,w]
        out = ops.upsample_bilinear(low.view(B, h, w, 1), int(img_size[0]), int(img_size[1]))
        return out.view(B, 1, int(img_size[0]), int(img_size[1]))


class MLPDecoder(MLPDecoderFM):
    """Decoders.py:239-286 (decoder_name='MLP'): the SegFormer-style head over the four encoder features only -- MLPDecoderFM
    without the main decoder's feature in the fuse conv (same parameter names, linear_fuse.0 takes 4*hidden channels)."""

    def __init__(self, in_channels, out_channel, hidden_channel=256, dropout_ratio=0.1, conv_norm=nn.BatchNorm2d):
        super().__init__(in_channels, out_channel, hidden_channel, outfeature_channel=0, dropout_ratio=dropout_ratio, conv_norm=conv_norm)


class _ASPP(nn.Module):
    """Utils/_deeplab.py:137-166: 1x1 conv | three dilated 3x3 convs | global-pool branch, each -> BN -> ReLU, concatenated and
    projected by a 1x1 conv -> BN -> ReLU -> Dropout(0.1).  Same child indices as the reference's Sequentials."""

    def __init__(self, in_channels, atrous_rates, out_channels=256):
        super().__init__()
        mods = [nn.Sequential(ConvParams(out_channels, in_channels, 1, 1, bias=False), BatchNormAct(out_channels, ACT_RELU), _NoParams())]
        for _ in atrous_rates:
            mods.append(nn.Sequential(ConvParams(out_channels, in_channels, 3, 3, bias=False), BatchNormAct(out_channels, ACT_RELU), _NoParams()))
        mods.append(nn.Sequential(_NoParams(), ConvParams(out_channels, in_channels, 1, 1, bias=False), BatchNormAct(out_channels, ACT_RELU), _NoParams()))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(ConvParams(out_channels, (len(atrous_rates) + 2) * out_channels, 1, 1, bias=False),
                                     BatchNormAct(out_channels, ACT_RELU), _NoParams(), nn.Dropout(0.1))
        self.rates = tuple(atrous_rates)
        self.out_channels = out_channels

    def forward(self, x):
        B, h, w, _ = x.shape
        oc = self.out_channels
        # five consumers of one tensor: behind ops.fork their gradients are added by OUR node, on whatever stream the sweep runs on (autograd's own accumulation
        # is launched on the forward's stream -- unordered with an aux sweep that runs on a stream of its own)
        xs = ops.fork(x, len(self.rates) + 2)
        res = [self.convs[0][1](ops.linear(xs[0], self.convs[0][0].weight.view(oc, -1)))]
        for i, r in enumerate(self.rates, start=1):
            res.append(self.convs[i][1](ops.conv3x3_dense(xs[i], self.convs[i][0].weight, None, 1, dilation=r)))
        pool = self.convs[-1]
        pooled = ops.linear(ops.global_avg_pool(xs[-1]), pool[1].weight.view(oc, -1)).view(B, 1, 1, oc)
        res.append(ops.upsample_bilinear(pool[2](pooled), h, w))                    # a 1x1 source: the constant, broadcast
        y = self.project[1](ops.linear(torch.cat(res, dim=-1), self.project[0].weight.view(oc, -1)))
        return ops.dropout(y, self.project[3].p, self.training)


class DeepLabV3Decoder(nn.Module):
    """Decoders.py:218-236 (decoder_name='DeepLabV3'): ASPP over the last encoder feature -> conv3x3 -> BN -> ReLU -> conv1x1
    to one channel -> bilinear upsample to the image (the 1x1 conv commutes with the upsample and runs at H/32)."""

    def __init__(self, in_channel, out_channel, aspp_dilate=(6, 12, 18), conv_norm=nn.BatchNorm2d):
        super().__init__()
        _check_norm(conv_norm)
        assert out_channel == 1
        self.classifier = nn.Sequential(_ASPP(in_channel, aspp_dilate), ConvParams(256, 256, 3, 3, bias=False), BatchNormAct(256, ACT_RELU),
                                        _NoParams(), ConvParams(out_channel, 256, 1, 1))

    def forward(self, features, img_size, out_feat=False):
        x = features[3] if isinstance(features, (list, tuple)) else features       # encoder_outs[-1] (mdvit.py:715-724 pass the 4 encoder features)
        c = self.classifier
        y = ops.conv3x3_dense(c[0](x), c[1].weight, None, 1)
        B, h, w, _ = y.shape
        bn = c[2]
        low = ops.bn_act_rowdot(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training, bn.act, c[4].weight, c[4].bias,
                                bn.eps, bn.momentum, 0.0)
        out = ops.upsample_bilinear(low.view(B, h, w, 1), int(img_size[0]), int(img_size[1]))
        return out.view(B, 1, int(img_size[0]), int(img_size[1]))
