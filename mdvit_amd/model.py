"""MDViT and BASE with the reference's constructor / forward / state_dict surface, computing on
hand-written HIP kernels (mdvit_amd.ops -> libmdvit_hip.so).

  MDViT <- Models/Transformer/mdvit.py:474-730      BASE <- Models/Transformer/base.py:340-512
Call surface kept (multi_train_MDViT.py:57-60,140-146; multi_train_BASE.py:66-68,168):
  MDViT(img_size=..., drop_rate=0.1, drop_path_rate=0.1, conv_norm=nn.BatchNorm2d, adapt_method='Sup',
        num_domains=4, decoder_name='MLPFM');  out, aux = model(img, domain_label, d)  |  model(img, d=d)
  BASE(drop_rate, drop_path_rate, conv_norm, adapt_method);  out = model(img)
Inputs NCHW fp32 on the GPU; outputs (B,1,H,W) fp32 logits.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from . import ops
from ._lib import ACT_RELU
from .blocks import (droppath_pool, dsn_domain, BatchNormAct, Conv2d_BN, ConvParams, DWCPatchEmbed, MHSA_stage_adapt, _NoParams, _check_norm,
                     init_weights_)
from .decode import DeepLabV3Decoder, MLPDecoder, MLPDecoderFM, UnetDecodingBlockTransformer

_FORK_GROUPS = os.environ.get("MDVIT_FORK_GROUPS", "1") != "0"      # the peer heads' feature copies as batch-group views of the fork itself (0: fork + split_groups, A/B)


class _EncoderDecoder(nn.Module):
    _base_semantics = False
    _dsn = 0            # > 0: domain-specific norms (MDViT_DSN): that many norms per BatchNorm / LayerNorm site

    def _build_trunk(self, img_size, in_chans, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale,
                     drop_rate, attn_drop_rate, drop_path_rate, norm_layer, conv_norm, adapt_method, num_domains, dsn=0):
        _check_norm(conv_norm)
        if num_stages != 4:
            raise NotImplementedError("num_stages must be 4")
        self.num_stages = num_stages
        self._dsn = dsn
        E = list(embed_dims)
        s1 = Conv2d_BN(in_chans, E[0] // 2, kernel_size=3, stride=2, pad=1, act_layer=nn.Hardswish, from_image=True, dsn=dsn)
        s2 = Conv2d_BN(E[0] // 2, E[0], kernel_size=3, stride=2, pad=1, act_layer=nn.Hardswish, dsn=dsn)
        if dsn:
            self.stem_1, self.stem_2 = s1, s2            # mdvit.py:773-790
        else:
            self.stem = nn.Sequential(s1, s2)
        self.patch_embed_stages = nn.ModuleList([
            DWCPatchEmbed(in_chans=E[i] if i == 0 else E[i - 1], embed_dim=E[i], patch_size=3, stride=1 if i == 0 else 2, conv_norm=conv_norm,
                          dsn=dsn)
            for i in range(num_stages)])

        def stage(i):
            return MHSA_stage_adapt((img_size // 2 ** (i + 2)) ** 2, E[i], num_layers=num_layers[i], num_heads=num_heads[i],
                                    mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias, qk_scale=qk_scale, drop_rate=drop_rate,
                                    attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
                                    adapt_method=adapt_method, num_domains=num_domains, base_semantics=self._base_semantics, dsn=dsn)

        self.mhsa_stages = nn.ModuleList([stage(i) for i in range(num_stages)])
        # the first adapter of the network in forward order (None without adapters): see ops.aux_stop / factor_att(aux_first=)
        att0 = self.mhsa_stages[0].mhca_blks[0].factoratt_crpe
        self._has_aux_first = hasattr(att0, "domain_layer")              # (a plain flag: a module reference would enter the state_dict)
        if self._has_aux_first:
            att0.aux_first = True
        if dsn:                                                          # mdvit.py:815-820
            self.bridge_conv1 = ConvParams(E[3], E[3], 3, 3)
            self.bridge_norms1 = nn.ModuleList([BatchNormAct(E[3], ACT_RELU) for _ in range(dsn)])
            self.bridge_conv2 = ConvParams(E[3] * 2, E[3], 3, 3)
            self.bridge_norms2 = nn.ModuleList([BatchNormAct(E[3] * 2, ACT_RELU) for _ in range(dsn)])
        else:
            self.bridge = nn.Sequential(ConvParams(E[3], E[3], 3, 3), BatchNormAct(E[3], ACT_RELU), _NoParams(),
                                        ConvParams(E[3] * 2, E[3], 3, 3), BatchNormAct(E[3] * 2, ACT_RELU), _NoParams())
        self.mhsa_list = [stage(i) for i in range(num_stages)]          # plain list, as in the reference (mdvit.py:568)
        self.decoder1 = UnetDecodingBlockTransformer(E[3] * 2, E[3], self.mhsa_list[3], conv_norm=conv_norm, dsn=dsn)
        self.decoder2 = UnetDecodingBlockTransformer(E[3], E[2], self.mhsa_list[2], conv_norm=conv_norm, dsn=dsn)
        self.decoder3 = UnetDecodingBlockTransformer(E[2], E[1], self.mhsa_list[1], conv_norm=conv_norm, dsn=dsn)
        self.decoder4 = UnetDecodingBlockTransformer(E[1], E[0], self.mhsa_list[0], conv_norm=conv_norm, dsn=dsn)
        self.finalconv = nn.Sequential(ConvParams(1, E[0], 1, 1))

    def _trunk(self, x, domain_label, groups: int = 1, split_for_heads: bool = False):
        """x NCHW image -> (logits (B,1,H,W), encoder_outs NHWC list, decoder4 output NHWC, image size, bridge output NHWC).
        groups > 1: x is `groups` equal consecutive domain batches; BatchNorm statistics stay per domain batch."""
        blocks = [blk for st in list(self.mhsa_stages) + list(self.mhsa_list) for blk in st.mhca_blks]
        keep = 1.0 - blocks[0].drop_path_p
        uniform = all(blk.drop_path_p == blocks[0].drop_path_p for blk in blocks)
        # every block's domain adapter (mdvit.py:272-276,301-303: a function of the labels and its own weights) in ONE launch at the top of the forward
        adapters = []
        if domain_label is not None:
            for blk in blocks:
                att = blk.factoratt_crpe
                if hasattr(att, "domain_layer"):
                    d0, d2 = att.domain_layer[0], att.domain_layer[2]
                    adapters.append((d0.weight, d0.bias, d2.weight, d2.bias, att.num_heads))
        with ops.bn_groups(groups), droppath_pool(len(blocks), x.shape[0], keep, x.device, enabled=self.training and uniform), ops.da_precomputed(domain_label, adapters):
            return self._trunk_impl(x, domain_label, groups if split_for_heads else 1)

    def _trunk_impl(self, x, domain_label, head_groups: int = 1):
        """head_groups > 1: the peer-head copies of the stage outputs / the decoder output come back as tuples of `head_groups` batch-group views whose gradients
        are summed with the trunk's in one pass (ops.fork_groups), not as whole tensors"""
        if x.dim() != 4:
            raise ValueError("expected a (B,C,H,W) image batch")
        B, _, Hi, Wi = x.shape
        x = self.stem_2(self.stem_1(x.float())) if self._dsn else self.stem[1](self.stem[0](x.float()))
        enc, skip = [], []          # every stage output has three consumers: the next stage / bridge, the decoder's skip path, the peer
        for idx in range(self.num_stages):          # heads -- forked explicitly so that the gradients are summed by ops._Fork, not by autograd
            x = self.patch_embed_stages[idx](x)
            _, H, W, Cn = x.shape
            if idx == 0 and self._has_aux_first:
                x = ops.aux_stop(x)          # the aux (data-gradient-only) sweep ends at the first adapter: the stem / patch embed carry none
            x = self.mhsa_stages[idx](x.view(B, H * W, Cn), H, W, domain_label).view(B, H, W, Cn)
            if head_groups > 1:
                x, s_, e_ = ops.fork_groups(x, 2, head_groups)
            else:
                x, s_, e_ = ops.fork(x, 3)
            skip.append(s_); enc.append(e_)
        if self._dsn:
            from .blocks import _bank_select
            out = _bank_select(self.bridge_norms1)(ops.conv3x3_dense(x, self.bridge_conv1.weight, self.bridge_conv1.bias, 1))
            out = _bank_select(self.bridge_norms2)(ops.conv3x3_dense(out, self.bridge_conv2.weight, self.bridge_conv2.bias, 1))
        else:
            out = ops.conv3x3_dense(x, self.bridge[0].weight, self.bridge[0].bias, 1)
            out = self.bridge[1](out)
            out = ops.conv3x3_dense(out, self.bridge[3].weight, self.bridge[3].bias, 1)
            out = self.bridge[4](out)
        if getattr(self, "decoder_name", None) == "Transformer":
            out, bridge_out = ops.fork(out, 2)          # the per-domain transformer peers start from the bridge output too
        else:
            bridge_out = out
        out = self.decoder1(out, skip[3], domain_label)
        out = self.decoder2(out, skip[2], domain_label)
        out = self.decoder3(out, skip[1], domain_label)
        out = self.decoder4(out, skip[0], domain_label)
        if head_groups > 1:
            out, dec4 = ops.fork_groups(out, 1, head_groups)
        else:
            out, dec4 = ops.fork(out, 2)                 # the final 1x1 conv and the peer heads' fifth feature
        _, h, w, _ = out.shape
        low = ops.rowdot(out, self.finalconv[0].weight, self.finalconv[0].bias)          # 1x1 conv (1 channel) at H/4
        logits = ops.upsample_bilinear(low.view(B, h, w, 1), Hi, Wi).view(B, 1, Hi, Wi)
        return logits, enc, dec4, (Hi, Wi), bridge_out

    @staticmethod
    def _pooled_feat(enc3):
        """adaptive_avg_pool2d(encoder_outs[3], 1) -- token mean via the column-sum kernel (not differentiable)."""
        B, H, W, Cn = enc3.shape
        feat = torch.empty((B, Cn), device=enc3.device, dtype=torch.float32)
        with torch.no_grad():
            wsp, wsb, _keep = ops._partials_ws(Cn, enc3.device)
            for b in range(B):
                ops.call("mdvit_colsum_f32", ops._p(enc3[b]), Cn, ops._p(feat[b]), None, wsp, wsb, H * W, Cn, 0.0, 0, 0, None, 1, 0, None, ops._stream())
        return feat / float(H * W)


class MDViT(_EncoderDecoder):
    def __init__(self, img_size=512, in_chans=3, num_stages=4, num_layers=[2, 2, 2, 2], embed_dims=[64, 128, 320, 512],
                 mlp_ratios=[8, 8, 4, 4], num_heads=[8, 8, 8, 8], qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=None, conv_norm=nn.BatchNorm2d, adapt_method=None, num_domains=4,
                 decoder_name="MLPFM", **kwargs):
        super().__init__()
        self.decoder_name = decoder_name
        self.adapt_method = adapt_method
        self._build_trunk(img_size, in_chans, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale,
                          drop_rate, attn_drop_rate, drop_path_rate, norm_layer, conv_norm, adapt_method, num_domains)
        self._build_peer_heads(embed_dims, decoder_name)
        if decoder_name == "Transformer":
            self._build_transformer_peers(img_size, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale, drop_rate,
                                          attn_drop_rate, drop_path_rate, norm_layer, conv_norm, num_domains)
        init_weights_(self)

    def _build_peer_heads(self, embed_dims, decoder_name):
        """mdvit.py:593-612 / 852-873: four peer heads -- 'MLPFM' (also fed the main decoder's last feature), 'MLP', 'DeepLabV3'"""
        if decoder_name == "MLPFM":
            mk = lambda: MLPDecoderFM(embed_dims, 1, 512)
        elif decoder_name == "MLP":
            mk = lambda: MLPDecoder(embed_dims, 1, 512)
        elif decoder_name == "DeepLabV3":
            mk = lambda: DeepLabV3Decoder(embed_dims[3], 1)
        elif decoder_name == "Transformer":
            return                                     # built by _build_transformer_peers (needs the trunk's hyper-parameters)
        else:
            raise ValueError(f"decoder_name={decoder_name!r}: the reference knows 'MLPFM', 'MLP', 'DeepLabV3', 'Transformer'")
        self.debranch1, self.debranch2, self.debranch3, self.debranch4 = mk(), mk(), mk(), mk()

    def _build_transformer_peers(self, img_size, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale, drop_rate,
                                 attn_drop_rate, drop_path_rate, norm_layer, conv_norm, num_domains):
        """mdvit.py:614-642: per domain a transformer decoder of its own -- four UnetDecodingBlockTransformer over MHSA stages
        WITHOUT Domain Adapter (adapt_method=False) and a 1x1 conv to one channel: debranchs.{d}.{0..3}, debranchs.{d}.4.0"""
        E = embed_dims
        peers = []
        for _ in range(num_domains):
            st = [MHSA_stage_adapt((img_size // 2 ** (i + 2)) ** 2, E[i], num_layers=num_layers[i], num_heads=num_heads[i], mlp_ratio=mlp_ratios[i],
                                   qkv_bias=qkv_bias, qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                                   drop_path_rate=drop_path_rate, norm_layer=norm_layer, adapt_method=False, num_domains=num_domains)
                  for i in range(num_stages)]
            peers.append(nn.ModuleList([UnetDecodingBlockTransformer(E[3] * 2, E[3], st[3], conv_norm=conv_norm),
                                        UnetDecodingBlockTransformer(E[3], E[2], st[2], conv_norm=conv_norm),
                                        UnetDecodingBlockTransformer(E[2], E[1], st[1], conv_norm=conv_norm),
                                        UnetDecodingBlockTransformer(E[1], E[0], st[0], conv_norm=conv_norm),
                                        nn.Sequential(ConvParams(1, E[0], 1, 1))]))
        self.debranchs = nn.ModuleList(peers)

    _HEADS = {"0": "debranch1", "1": "debranch2", "2": "debranch3", "3": "debranch4"}

    def _compose_peers(self, ds):
        """the composed 1x1-conv weights of the MLP-style heads of domains ds in grouped launches (MLPDecoderFM.compose_many), or None per head"""
        if self.decoder_name not in ("MLPFM", "MLP") or any(dd not in self._HEADS for dd in ds) or len(set(ds)) != len(ds):
            return [None] * len(ds)
        return MLPDecoderFM.compose_many([getattr(self, self._HEADS[dd]) for dd in ds])

    def _peer_out(self, dd, feats, bridge_out, img_size, composed=None):
        """the peer head of domain id string dd on (a batch group of) the trunk's features"""
        if self.decoder_name == "Transformer":
            peer = self.debranchs[int(dd)]
            a = bridge_out
            for j in range(4):
                a = peer[j](a, feats[3 - j])                       # no domain label: these decoders carry no adapter
            B, h, w, _ = a.shape
            low = ops.rowdot(a, peer[4][0].weight, peer[4][0].bias)                        # 1x1 conv at H/4, then upsample
            return ops.upsample_bilinear(low.view(B, h, w, 1), int(img_size[0]), int(img_size[1])).view(B, 1, int(img_size[0]), int(img_size[1]))
        heads = self._HEADS
        if dd not in heads:
            return None
        if composed is not None:
            return getattr(self, heads[dd])(feats, img_size=img_size, composed=composed)
        return getattr(self, heads[dd])(feats, img_size=img_size)

    def forward(self, x, domain_label=None, d=None, out_feat=False, out_seg=True):
        """d: the reference's domain id string ('0'..'3').  Extension: a list/tuple of G domain ids runs a DOMAIN-BATCHED
        forward -- x (and domain_label) hold G equal consecutive domain batches; BatchNorm statistics are kept per
        domain batch and each batch goes through its own peer head, so the result equals G separate forwards
        (multi_train_MDViT.py:137-153) concatenated along the batch axis."""
        if isinstance(d, (list, tuple)):
            return self._forward_domains(x, domain_label, [str(v) for v in d], out_feat, out_seg)
        logits, enc, dec4, img_size, bridge_out = self._trunk(x, domain_label)
        if not out_seg:
            return {"seg": None, "feat": self._pooled_feat(enc[3])}
        aux_out = self._peer_out(d, enc + [dec4], bridge_out, img_size)
        if out_feat:
            return {"seg": [logits, aux_out], "feat": self._pooled_feat(enc[3])}
        return [logits, aux_out]


    def _peer_heads(self, ds, parts, bparts, img_size):
        """The G peer heads of a domain-batched forward.  They share nothing (each reads its own batch group of the trunk's
        features through its own weights), so each runs on a stream of its own -- and so does its backward: autograd executes a
        node's backward on the stream its forward ran on and orders the streams at the graph's edges."""
        G = len(ds)
        streams = [ops.peer_stream(g, G * parts[0][0].shape[0]) for g in range(G)] if (G > 1 and parts[0][0].is_cuda) else [None] * G
        if any(s is None for s in streams):
            comp = self._compose_peers(ds)             # G x 4 weight compositions: grouped launches, ahead of the heads
            if self.decoder_name in ("MLPFM", "MLP") and all(c is not None for c in comp):
                comp = MLPDecoderFM.grouped_lows(comp, [[pf[g] for pf in parts] for g in range(G)])
            return [self._peer_out(dd, [pf[g] for pf in parts], bparts[g], img_size, comp[g]) for g, dd in enumerate(ds)]
        main = torch.cuda.current_stream()
        aux = []
        for g, dd in enumerate(ds):
            st = streams[g]
            st.wait_stream(main)                                   # the trunk's features are complete
            feats = [pf[g] for pf in parts]
            for t in feats + ([bparts[g]] if bparts[g] is not None else []):
                t.record_stream(st)                                # main-stream memory read on st: keep it until st is done with it
            with torch.cuda.stream(st):
                a = self._peer_out(dd, feats, bparts[g], img_size)
            if a is not None:
                a.record_stream(main)                              # consumed (concatenated, losses) on the main stream
            aux.append(a)
        for st in streams:
            main.wait_stream(st)
        return aux

    def _forward_domains(self, x, domain_label, ds, out_feat, out_seg):
        G = len(ds)
        if x.shape[0] % G:
            raise ValueError(f"batch {x.shape[0]} is not {G} equal domain batches")
        if not out_seg:
            logits, enc, dec4, img_size, bridge_out = self._trunk(x, domain_label, groups=G)
            return {"seg": None, "feat": self._pooled_feat(enc[3])}
        logits, enc, dec4, img_size, bridge_out = self._trunk(x, domain_label, groups=G, split_for_heads=G > 1 and _FORK_GROUPS)
        parts = [f if isinstance(f, tuple) else ops.split_groups(f, G) for f in enc + [dec4]]          # per feature: G batch views
        bparts = ops.split_groups(bridge_out, G) if self.decoder_name == "Transformer" else [None] * G
        aux = self._peer_heads(ds, parts, bparts, img_size)
        aux_out = None if any(a is None for a in aux) else torch.cat(aux, 0)
        if out_feat:
            e3 = enc[3] if not isinstance(enc[3], tuple) else torch.cat([t.detach() for t in enc[3]], 0)
            return {"seg": [logits, aux_out], "feat": self._pooled_feat(e3)}
        return [logits, aux_out]


class MDViT_DSN(MDViT):
    """mdvit.py:735-960: MDViT with domain-specific norms -- every trunk BatchNorm / LayerNorm is a ModuleList of
    num_domains norms (stem_{1,2}.bns, patch_conv.bns, norm1s / norm2s, bridge_norms{1,2}, conv_after.bns) indexed by
    int(d); everything else, the peer heads included, is MDViT.  `d` is therefore required.  A list of DISTINCT ids runs
    the domain-batched forward: norm d_g on the g-th batch group, through the group-indexed BN / LN kernels."""

    def __init__(self, img_size=512, in_chans=3, num_stages=4, num_layers=[2, 2, 2, 2], embed_dims=[64, 128, 320, 512],
                 mlp_ratios=[8, 8, 4, 4], num_heads=[8, 8, 8, 8], qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=None, conv_norm=nn.BatchNorm2d, adapt_method=None, num_domains=4,
                 decoder_name="MLP", **kwargs):
        _EncoderDecoder.__init__(self)
        if decoder_name == "Transformer":
            raise NotImplementedError("MDViT_DSN has no 'Transformer' peer decoders (mdvit.py:852-873 builds MLP / DeepLabV3 / MLPFM only)")
        self.decoder_name = decoder_name
        self.adapt_method = adapt_method
        self._build_trunk(img_size, in_chans, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale,
                          drop_rate, attn_drop_rate, drop_path_rate, norm_layer, conv_norm, adapt_method, num_domains, dsn=num_domains)
        self._build_peer_heads(embed_dims, decoder_name)
        init_weights_(self)

    def forward(self, x, domain_label=None, d=None, out_feat=False, out_seg=True):
        if isinstance(d, (list, tuple)):
            with dsn_domain(tuple(int(v) for v in d)):
                return self._forward_domains(x, domain_label, [str(v) for v in d], out_feat, out_seg)
        with dsn_domain(int(d)):
            return MDViT.forward(self, x, domain_label, str(d), out_feat, out_seg)


class BASE(_EncoderDecoder):
    _base_semantics = True          # base.py:216 applies the adapter whenever a domain_label is given

    def __init__(self, img_size=512, in_chans=3, num_stages=4, num_layers=[2, 2, 2, 2], embed_dims=[64, 128, 320, 512],
                 mlp_ratios=[8, 8, 4, 4], num_heads=[8, 8, 8, 8], qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=None, conv_norm=nn.BatchNorm2d, adapt_method=None, num_domains=4, **kwargs):
        super().__init__()
        self.adapt_method = adapt_method
        self._build_trunk(img_size, in_chans, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale,
                          drop_rate, attn_drop_rate, drop_path_rate, norm_layer, conv_norm, adapt_method, num_domains)
        init_weights_(self)

    def forward(self, x, domain_label=None, out_feat=False, out_seg=True):
        logits, enc, dec4, _, _ = self._trunk(x, domain_label)
        if not out_seg:
            return {"seg": None, "feat": self._pooled_feat(enc[3])}
        if out_feat:
            return {"seg": logits, "feat": self._pooled_feat(enc[3])}
        return logits


class BASE_DSN(BASE):
    """base.py:515-700: BASE with domain-specific norms (the trunk of MDViT_DSN: stem_{1,2}.bns, norm1s / norm2s, bridge_norms{1,2},
    conv_after.bns, indexed by int(d)); forward(x, domain_label, d) -> logits.  A list of distinct ids runs the domain-batched
    forward through the group-indexed BN / LN kernels."""

    def __init__(self, img_size=512, in_chans=3, num_stages=4, num_layers=[2, 2, 2, 2], embed_dims=[64, 128, 320, 512],
                 mlp_ratios=[8, 8, 4, 4], num_heads=[8, 8, 8, 8], qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=None, conv_norm=nn.BatchNorm2d, adapt_method=None, num_domains=4, **kwargs):
        _EncoderDecoder.__init__(self)
        self.adapt_method = adapt_method
        self._build_trunk(img_size, in_chans, num_stages, num_layers, embed_dims, mlp_ratios, num_heads, qkv_bias, qk_scale,
                          drop_rate, attn_drop_rate, drop_path_rate, norm_layer, conv_norm, adapt_method, num_domains, dsn=num_domains)
        init_weights_(self)

    def forward(self, x, domain_label=None, d=None, out_feat=False, out_seg=True):
        if isinstance(d, (list, tuple)):
            G = len(d)
            if x.shape[0] % G:
                raise ValueError(f"batch {x.shape[0]} is not {G} equal domain batches")
            with dsn_domain(tuple(int(v) for v in d)):
                logits, enc, _, _, _ = self._trunk(x, domain_label, groups=G)
        else:
            with dsn_domain(int(d)):
                logits, enc, _, _, _ = self._trunk(x, domain_label)
        if not out_seg:
            return {"seg": None, "feat": self._pooled_feat(enc[3])}
        if out_feat:
            return {"seg": logits, "feat": self._pooled_feat(enc[3])}
        return logits
