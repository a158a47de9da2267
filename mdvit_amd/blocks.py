"""Building blocks of the MDViT path as nn.Modules whose forward runs HIP kernels (mdvit_amd.ops).

Class names, constructor arguments, parameter names and shapes mirror the reference so its
state_dict loads unchanged:
  Conv2d_BN, ConvPosEnc, ConvRelPosEnc, Mlp, FactorAtt_ConvRelPosEnc   <- Models/Transformer/mpvit.py
  DWConv2d_BN, DWCPatchEmbed, FactorAtt_ConvRelPosEnc_Sup, SerialBlock_adapt, MHSA_stage_adapt
                                                                      <- Models/Transformer/mdvit.py
Internally every activation is NHWC: images [B,H,W,C], tokens [B,H*W,C] (the same memory).
Parameters live in plain containers (no torch forward exists to fall back to).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn

from . import ops
from ._lib import ACT_HSWISH, ACT_NONE, ACT_RELU


# ---- parameter containers (state_dict-compatible with nn.Conv2d / nn.Linear / nn.LayerNorm / nn.BatchNorm2d)
class ConvParams(nn.Module):
    def __init__(self, out_ch, in_per_group, kh, kw, bias=True, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_ch, in_per_group, kh, kw))
        self.bias = nn.Parameter(torch.empty(out_ch)) if bias else None
        self.kernel_size, self.out_channels, self.groups = (kh, kw), out_ch, groups

    def forward(self, *a, **k):
        raise RuntimeError("ConvParams only holds parameters; compute goes through mdvit_amd.ops (HIP)")


class LinearParams(nn.Module):
    def __init__(self, in_f, out_f, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.empty(out_f)) if bias else None

    def forward(self, *a, **k):
        raise RuntimeError("LinearParams only holds parameters; compute goes through mdvit_amd.ops (HIP)")


class LayerNormParams(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def fork(self, x):
        """(LN(x), x) with the residual-branch gradient folded into the LayerNorm backward (ops._LayerNorm)."""
        return ops.layer_norm_fork(x, self.weight, self.bias, self.eps)


class BatchNormAct(nn.Module):
    """nn.BatchNorm2d state (weight, bias, running_mean, running_var, num_batches_tracked) + fused activation."""

    def __init__(self, ch, act=ACT_NONE, eps=1e-5, momentum=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.register_buffer("running_mean", torch.zeros(ch))
        self.register_buffer("running_var", torch.ones(ch))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.act, self.eps, self.momentum = act, eps, momentum

    def forward(self, y, drop2d_p: float = 0.0):
        return ops.bn_act(y, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                          self.training, self.act, self.eps, self.momentum, drop2d_p if self.training else 0.0)


def _check_norm(conv_norm):
    if conv_norm is not nn.BatchNorm2d:
        raise NotImplementedError("only conv_norm=nn.BatchNorm2d is built (the reference's train scripts use nothing else)")


class _NoParams(nn.Module):
    """placeholder keeping nn.Sequential indices aligned with the reference (ReLU / Identity slots)."""

    def forward(self, x):
        return x


# ---- domain-specific norms (the *_M classes of the reference: mdvit.py:23-70,127-179,364-412; Decoders.py:66-118) ------
# A module built with dsn=D holds D norms in a ModuleList under the reference's name (bns / norm1s / norm2s) and applies
# the one the model's forward selected with dsn_domain(int(d)); dsn=0 is the ordinary single-norm module.
# dsn_domain((d0, d1, ...)) -- a tuple, one DISTINCT domain per group of a domain-batched forward (ops.bn_groups) --
# applies norm d_g to the g-th batch group in ONE launch: the bank's parameters are stacked to [G, C] rows and the
# BN / LN kernels index the row by group (include/mdvit_hip.h: per_group_affine / groups).
_dsn_domain = None


class _BankBN:
    """G BatchNormAct modules applied to the G batch groups of y in one pass."""

    def __init__(self, mods):
        self.mods = mods

    def __call__(self, y, drop2d_p: float = 0.0):
        mods, m0 = self.mods, self.mods[0]
        gamma = torch.stack([m.weight for m in mods])
        beta = torch.stack([m.bias for m in mods])
        with torch.no_grad():
            rm = torch.stack([m.running_mean for m in mods])
            rv = torch.stack([m.running_var for m in mods])
        z = ops.bn_act(y, gamma, beta, rm, rv, None, m0.training, m0.act, m0.eps, m0.momentum, drop2d_p if m0.training else 0.0)
        if m0.training:
            with torch.no_grad():              # hand the updated rows back to the modules' own buffers (state_dict layout)
                torch._foreach_copy_([m.running_mean for m in mods] + [m.running_var for m in mods], list(rm.unbind(0)) + list(rv.unbind(0)))
                torch._foreach_add_([m.num_batches_tracked for m in mods], 1)
        return z


class _BankLN:
    def __init__(self, mods):
        self.mods = mods

    def _params(self):
        return torch.stack([m.weight for m in self.mods]), torch.stack([m.bias for m in self.mods])

    def __call__(self, x):
        g, b = self._params()
        return ops.layer_norm(x, g, b, self.mods[0].eps)

    def fork(self, x):
        g, b = self._params()
        return ops.layer_norm_fork(x, g, b, self.mods[0].eps)


def _bank_select(b):
    """the norm (or group-batched bank view) the active dsn_domain selects from ModuleList b"""
    d = _dsn_domain
    if isinstance(d, tuple):
        if len(set(d)) != len(d) or not all(0 <= v < len(b) for v in d):
            raise ValueError(f"domain-batched domain-specific norms need distinct domain ids in 0..{len(b) - 1} (got {d!r})")
        if len(d) != ops._bn_groups:
            raise ValueError(f"{len(d)} domain ids for bn_groups({ops._bn_groups})")
        mods = [b[v] for v in d]
        return _BankBN(mods) if isinstance(mods[0], BatchNormAct) else _BankLN(mods)
    if d is None or not 0 <= d < len(b):
        raise ValueError(f"domain-specific norms need the domain id d in 0..{len(b) - 1} (got {d!r})")
    return b[d]


class dsn_domain:
    def __init__(self, d):
        self.d = d

    def __enter__(self):
        global _dsn_domain
        self.prev, _dsn_domain = _dsn_domain, self.d
        return self

    def __exit__(self, *exc):
        global _dsn_domain
        _dsn_domain = self.prev
        return False


def _bank(make, dsn):
    return nn.ModuleList([make() for _ in range(dsn)])


def _pick(mod, single: str, bank: str):
    b = getattr(mod, bank, None)
    if b is None:
        return getattr(mod, single)
    return _bank_select(b)


# ---- conv blocks -----------------------------------------------------------------------------------
class Conv2d_BN(nn.Module):
    """stem conv: 3x3 s2 p1 (no bias) -> BN -> Hardswish (mpvit.py:81-124).  First stem conv reads the
    NCHW image directly; the second goes im2col + MFMA GEMM."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=2, pad=1, act_layer=nn.Hardswish, norm_layer=nn.BatchNorm2d, from_image=False,
                 dsn=0):
        super().__init__()
        _check_norm(norm_layer)
        assert kernel_size == 3 and pad == 1
        self.conv = ConvParams(out_ch, in_ch, 3, 3, bias=False)
        mk = lambda: BatchNormAct(out_ch, ACT_HSWISH if act_layer is nn.Hardswish else ACT_NONE)
        if dsn:
            self.bns = _bank(mk, dsn)
        else:
            self.bn = mk()
        self.stride, self.from_image = stride, from_image

    def forward(self, x):
        if self.from_image:
            assert self.stride == 2
            y = ops.stem_conv(x, self.conv.weight)
        else:
            y = ops.conv3x3_dense(x, self.conv.weight, None, self.stride)
        return _pick(self, "bn", "bns")(y)


class DWConv2d_BN(nn.Module):
    """dw3x3(groups=in) -> pw1x1 -> BN -> Hardswish (mdvit.py:74-123)."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, norm_layer=nn.BatchNorm2d, act_layer=nn.Hardswish, dsn=0):
        super().__init__()
        _check_norm(norm_layer)
        assert kernel_size == 3
        self.dwconv = ConvParams(in_ch, 1, 3, 3, bias=False, groups=in_ch)
        self.pwconv = ConvParams(out_ch, in_ch, 1, 1, bias=False)
        if dsn:
            self.bns = _bank(lambda: BatchNormAct(out_ch, ACT_HSWISH), dsn)
        else:
            self.bn = BatchNormAct(out_ch, ACT_HSWISH)
        self.stride = stride

    def forward(self, x):
        t = ops.dwconv3x3(x, self.dwconv.weight, None, self.stride, False)
        return _pick(self, "bn", "bns")(ops.linear(t, self.pwconv.weight))


class DWCPatchEmbed(nn.Module):
    """mdvit.py:183-208."""

    def __init__(self, in_chans=3, embed_dim=768, patch_size=16, stride=1, conv_norm=nn.BatchNorm2d, act_layer=nn.Hardswish, dsn=0):
        super().__init__()
        self.patch_conv = DWConv2d_BN(in_chans, embed_dim, kernel_size=patch_size, stride=stride, norm_layer=conv_norm, act_layer=act_layer,
                                      dsn=dsn)

    def forward(self, x):
        return self.patch_conv(x)


class DecoderDWConv2d_BN(nn.Module):
    """Decoders.py flavour (:15-63): Conv2d(in=2*out, out, 3, groups=out) -> pw(out,out) -> BN -> Hardswish,
    applied to cat(skip, up) without materialising the concat."""

    def __init__(self, in_ch, out_ch, norm_layer=nn.BatchNorm2d, dsn=0):
        super().__init__()
        _check_norm(norm_layer)
        assert in_ch == 2 * out_ch
        self.dwconv = ConvParams(out_ch, 2, 3, 3, bias=False, groups=out_ch)
        self.pwconv = ConvParams(out_ch, out_ch, 1, 1, bias=False)
        if dsn:
            self.bns = _bank(lambda: BatchNormAct(out_ch, ACT_HSWISH), dsn)
        else:
            self.bn = BatchNormAct(out_ch, ACT_HSWISH)

    def forward(self, skip, up):
        t = ops.gconv2_3x3(skip, up, self.dwconv.weight)
        return _pick(self, "bn", "bns")(ops.linear(t, self.pwconv.weight))


# ---- transformer block ------------------------------------------------------------------------------
class ConvPosEnc(nn.Module):
    """x + dwconv3x3_bias(x)  (mpvit.py:229-248)."""

    def __init__(self, dim, k=3):
        super().__init__()
        assert k == 3
        self.proj = ConvParams(dim, 1, 3, 3, bias=True, groups=dim)

    def forward(self, x, size: Tuple[int, int]):
        B, N, Cn = x.shape
        H, W = size
        return ops.dwconv3x3(x.view(B, H, W, Cn), self.proj.weight, self.proj.bias, 1, True).view(B, N, Cn)


class ConvRelPosEnc(nn.Module):
    """Holds the 3/5/7 depthwise windows (mpvit.py:251-318); applied inside ops.factor_att."""

    def __init__(self, Ch, h, window):
        super().__init__()
        if isinstance(window, int):
            window = {window: h}
        if sorted(window) != [3, 5, 7] or sum(window.values()) != h:
            raise NotImplementedError("the HIP attention kernel is built for the reference's {3:2, 5:3, 7:3}-style windows")
        self.window = dict(window)
        self.head_splits = [window[3], window[5], window[7]]
        self.conv_list = nn.ModuleList([ConvParams(window[k] * Ch, 1, k, k, bias=True, groups=window[k] * Ch) for k in (3, 5, 7)])

    def params(self):
        c = self.conv_list
        return (c[0].weight, c[0].bias, c[1].weight, c[1].bias, c[2].weight, c[2].bias)


class _FactorAttBase(nn.Module):
    def _attend(self, x, size, domain_label, da_params, res, rowscale):
        B, N, Cn = x.shape
        H, W = size
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        y = ops.factor_att(qkv, self.crpe.params(), H, W, self.num_heads, self.crpe.head_splits, domain_label, da_params,
                           aux_first=getattr(self, "aux_first", False))
        # proj + proj_drop (+ DropPath + residual when the caller hands them in)
        return ops.linear(y, self.proj.weight, self.proj.bias, residual=res, rowscale=rowscale,
                          drop_p=self.proj_drop_p if self.training else 0.0, rows_per_scale=N)


class FactorAtt_ConvRelPosEnc(_FactorAttBase):
    """mpvit.py:321-373 (no domain adapter)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, shared_crpe=None):
        super().__init__()
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not built")
        self.num_heads = num_heads
        self.qkv = LinearParams(dim, dim * 3, bias=qkv_bias)
        self.proj = LinearParams(dim, dim)
        self.proj_drop_p = proj_drop
        self.crpe = shared_crpe

    def forward(self, x, size, _res=None, _rowscale=None):
        return self._attend(x, size, None, None, _res, _rowscale)


class FactorAtt_ConvRelPosEnc_Sup(_FactorAttBase):
    """mdvit.py:243-313 (domain adapter: softmax over heads of MLP(one_hot))."""

    def __init__(self, seq_length, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 shared_crpe=None, r=2, num_domains=4):
        super().__init__()
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not built")
        self.num_heads = num_heads
        hidden = max(dim // r, 4)
        self.qkv = LinearParams(dim, dim * 3, bias=qkv_bias)
        self.proj = LinearParams(dim, dim)
        self.proj_drop_p = proj_drop
        self.domain_layer = nn.Sequential(LinearParams(num_domains, hidden), _NoParams(), LinearParams(hidden, dim))
        self.crpe = shared_crpe

    def forward(self, x, size, domain_label, _res=None, _rowscale=None):
        d0, d2 = self.domain_layer[0], self.domain_layer[2]
        return self._attend(x, size, domain_label, (d0.weight, d0.bias, d2.weight, d2.bias), _res, _rowscale)


class Mlp(nn.Module):
    """mpvit.py:51-78."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert out_features == in_features and act_layer is nn.GELU
        self.fc1 = LinearParams(in_features, hidden_features)
        self.fc2 = LinearParams(hidden_features, out_features)
        self.drop_p = drop

    def forward(self, x, _res=None, _rowscale=None):
        N = x.shape[1]
        res = _res if _res is not None else torch.zeros_like(x)
        return ops.mlp_residual(x, res, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, _rowscale,
                                self.drop_p if self.training else 0.0, N)


# DropPath masks of one forward, drawn at once by the model (model._trunk): {"m": [n_blocks,2,B], "keep", "B", "next"}
_droppath_pool = None


class droppath_pool:
    """with droppath_pool(n_blocks, B, keep, device): the SerialBlocks inside take their per-sample DropPath scales from one
    pre-drawn tensor (same distribution: an independent Bernoulli(keep)/keep per block, branch and sample)."""

    def __init__(self, n_blocks, B, keep, device, enabled=True):
        self.args = (n_blocks, B, keep, device, enabled)

    def __enter__(self):
        global _droppath_pool
        n, B, keep, device, enabled = self.args
        self.prev = _droppath_pool
        if enabled and keep < 1.0:
            m = ops.droppath_scales((n, 2, B), keep, device) if torch.device(device).type == "cuda" else (torch.rand((n, 2, B), device=device) < keep).float() / keep
            _droppath_pool = {"m": m, "keep": keep, "B": B, "next": 0}
        return self

    def __exit__(self, *exc):
        global _droppath_pool
        _droppath_pool = self.prev
        return False


class SerialBlock_adapt(nn.Module):
    """mdvit.py:316-361: cpe -> LN -> attention(+DA) -> DropPath+res -> LN -> MLP -> DropPath+res."""

    def __init__(self, seq_length, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=None, shared_cpe=None, shared_crpe=None, adapt_method=None,
                 num_domains=4, base_semantics=False, dsn=0):
        super().__init__()
        self.cpe = shared_cpe
        if dsn:
            self.norm1s = _bank(lambda: LayerNormParams(dim, 1e-6), dsn)
        else:
            self.norm1 = LayerNormParams(dim, 1e-6)
        self.adapt_method = adapt_method
        self.base_semantics = base_semantics
        if adapt_method == "Sup":
            self.factoratt_crpe = FactorAtt_ConvRelPosEnc_Sup(seq_length, dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                                              attn_drop=attn_drop, proj_drop=drop, shared_crpe=shared_crpe, num_domains=num_domains)
        else:
            self.factoratt_crpe = FactorAtt_ConvRelPosEnc(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                                          attn_drop=attn_drop, proj_drop=drop, shared_crpe=shared_crpe)
        self.drop_path_p = drop_path
        if dsn:
            self.norm2s = _bank(lambda: LayerNormParams(dim, 1e-6), dsn)
        else:
            self.norm2 = LayerNormParams(dim, 1e-6)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _droppath_scales(self, B, device):
        if not self.training or self.drop_path_p <= 0.0:
            return None, None
        keep = 1.0 - self.drop_path_p
        pool = _droppath_pool
        if pool is not None and pool["keep"] == keep and pool["B"] == B and pool["next"] < pool["m"].shape[0]:
            m = pool["m"][pool["next"]]            # pre-drawn for the whole forward (one torch.rand instead of one per block)
            pool["next"] += 1
            return m[0], m[1]
        m = ops.droppath_scales((2, B), keep, device)                      # per-sample masks for both branches
        return m[0], m[1]

    def _block_entry(self, x, size, domain_label, use_da):
        """the whole block as one C call (ops.serial_block), or None when the configuration needs the operator-level path"""
        att, mlp = self.factoratt_crpe, self.mlp
        if not x.is_cuda or att.proj_drop_p != mlp.drop_p or att.qkv.bias is None:
            return None
        if not torch.is_grad_enabled():
            return None        # inference: the block entry lays out the whole save buffer (~17 T C floats) at once; the operator path frees intermediates as it goes
        n1, n2 = _pick(self, "norm1", "norm1s"), _pick(self, "norm2", "norm2s")
        if isinstance(n1, _BankLN):
            (g1, b1), (g2, b2) = n1._params(), n2._params()
            groups, eps = g1.shape[0], n1.mods[0].eps
        else:
            g1, b1, g2, b2, groups, eps = n1.weight, n1.bias, n2.weight, n2.bias, 1, n1.eps
        da = [None] * 4
        if use_da:
            d0, d2 = att.domain_layer[0], att.domain_layer[2]
            da = [d0.weight, d0.bias, d2.weight, d2.bias]
        params = [self.cpe.proj.weight, self.cpe.proj.bias, g1, b1, att.qkv.weight, att.qkv.bias, *att.crpe.params(), *da, att.proj.weight, att.proj.bias,
                  g2, b2, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias]
        if not ops.block_entry_ok(x.shape[-1], mlp.fc1.weight.shape[0], params):
            return None
        s1, s2 = self._droppath_scales(x.shape[0], x.device)
        meta = (int(size[0]), int(size[1]), att.num_heads, tuple(att.crpe.head_splits), float(eps), float(mlp.drop_p if self.training else 0.0), int(groups),
                bool(getattr(att, "aux_first", False)))
        return ops.serial_block(x, domain_label if use_da else None, s1, s2, meta, params)

    def forward(self, x, size: Tuple[int, int], domain_label=None):
        use_da = (domain_label is not None) if self.base_semantics else (self.adapt_method is not None and domain_label is not None)
        if use_da and not isinstance(self.factoratt_crpe, FactorAtt_ConvRelPosEnc_Sup):
            raise TypeError("forward() got a domain_label but this block was built without adapt_method='Sup' "
                            "(the reference raises here too: mdvit.py:350-351)")
        if not use_da and isinstance(self.factoratt_crpe, FactorAtt_ConvRelPosEnc_Sup):
            raise TypeError("adapt_method='Sup' blocks need a domain_label (mdvit.py:281)")
        y = self._block_entry(x, size, domain_label, use_da)
        if y is not None:
            return y
        x = self.cpe(x, size)
        s1, s2 = self._droppath_scales(x.shape[0], x.device)
        cur, x = _pick(self, "norm1", "norm1s").fork(x)
        if use_da:
            x = self.factoratt_crpe(cur, size, domain_label, _res=x, _rowscale=s1)
        else:
            x = self.factoratt_crpe(cur, size, _res=x, _rowscale=s1)
        cur, x = _pick(self, "norm2", "norm2s").fork(x)
        return self.mlp(cur, _res=x, _rowscale=s2)


class MHSA_stage_adapt(nn.Module):
    """mdvit.py:415-440: one shared ConvPosEnc / ConvRelPosEnc per stage, num_layers serial blocks."""

    def __init__(self, seq_length, dim, num_layers, num_heads, mlp_ratio, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0, num_domains=4, norm_layer=None, adapt_method=None,
                 crpe_window={3: 2, 5: 3, 7: 3}, base_semantics=False, dsn=0):
        super().__init__()
        self.cpe = ConvPosEnc(dim, k=3)
        self.crpe = ConvRelPosEnc(Ch=dim // num_heads, h=num_heads, window=crpe_window)
        self.mhca_blks = nn.ModuleList([
            SerialBlock_adapt(seq_length, dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate, drop_path_rate,
                              nn.GELU, norm_layer, self.cpe, self.crpe, adapt_method, num_domains, base_semantics, dsn)
            for _ in range(num_layers)])

    def forward(self, input, H, W, domain_label=None):
        for blk in self.mhca_blks:
            input = blk(input, (H, W), domain_label)
        return input


def init_weights_(module: nn.Module):
    """The reference's `_init_weights` (mdvit.py:648-664): Linear trunc_normal(.02)/0, LayerNorm 1/0,
    Conv normal(0, sqrt(2/fan_out)) with fan_out = kh*kw*out/groups, BatchNorm 1/0."""
    for m in module.modules():
        if isinstance(m, LinearParams):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, LayerNormParams):
            nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
        elif isinstance(m, ConvParams):
            out_ch, in_pg, kh, kw = m.weight.shape
            fan_out = kh * kw * out_ch // m.groups
            nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, BatchNormAct):
            nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
