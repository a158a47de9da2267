"""CPU oracle of the TransFuse_S_adapt path (BASELINE configs[4]): a functional restatement in plain torch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by mdvit_amd.  Every function cites the reference lines it follows (paths under /root/reference/):
  Models/Hybrid_models/TransFuseFolder/TransFuse.py          TransFuse_S_adapt :182-283, BiFusion_block :25-76, Up :523-549,
                                                              Attention_block :552-576, DoubleConv :579-598, Residual :601-636, Conv :639-656
  Models/Hybrid_models/TransFuseFolder/vision_transformer.py  Attention_Sup :125-169, Block_adapt :195-214, Mlp :73-88, PatchEmbed :218-240
  Models/Hybrid_models/TransFuseFolder/DeiT.py                DeiT_adapt.forward :57-71, deit_small_patch16_224_adapt :116-139
  multi_train_TransFuse.py                                    structure_loss :29-38, step loss :162-172
The CNN branch is torchvision's ResNet-34 (not installed here, not under /root/reference): conv1 7x7/2 (no bias) - bn1 - relu -
maxpool 3x3/2 pad 1 - layer1 (3 BasicBlocks, 64) - layer2 (4, 128, first stride 2) - layer3 (6, 256, first stride 2); BasicBlock =
conv3x3(stride) - BN - ReLU - conv3x3 - BN - (+ x, or + BN(conv1x1 stride-s (x)) on the first block of layer2 / layer3) - ReLU,
all convolutions without bias, BN eps 1e-5 momentum 0.1 (He et al. 2016; torchvision.models.resnet).  layer4 / fc are replaced by
Identity in the reference (TransFuse.py:190-191) and hold no parameters.
Pinned by tests/golden/transfuse_step_256.npz, generated from the reference's own TransFuse_S_adapt (oracle/gen_golden.py,
oracle/ref_import.import_transfuse: the reference class on top of the ResNet-34 restatement above).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .params import uniform_pm1, _stream_id

Tensor = torch.Tensor
EMBED, DEPTH, HEADS, MLP_RATIO, PATCH, NUM_DOMAINS = 384, 8, 6, 4, 16, 4


# ---------------------------------------------------------------------------------------------------------------------
# parameter inventory (the reference's state_dict: 630 keys, 26.87 M parameters) and the build-owned generator
# ---------------------------------------------------------------------------------------------------------------------
def _bn(spec, prefix, c):
    spec[prefix + ".weight"] = ("bn_w", (c,)); spec[prefix + ".bias"] = ("bn_b", (c,))
    spec[prefix + ".running_mean"] = ("bn_rm", (c,)); spec[prefix + ".running_var"] = ("bn_rv", (c,))
    spec[prefix + ".num_batches_tracked"] = ("bn_nbt", ())


def _conv(spec, prefix, out_c, in_c, k, bias=True):
    spec[prefix + ".weight"] = ("conv", (out_c, in_c, k, k))
    if bias:
        spec[prefix + ".bias"] = ("bias", (out_c,))


def _Conv(spec, prefix, inp, out, k, bn=False, bias=True):          # TransFuse.py:639-656
    _conv(spec, prefix + ".conv", out, inp, k, bias)
    if bn:
        _bn(spec, prefix + ".bn", out)


def _double_conv(spec, prefix, inp, out):                          # TransFuse.py:579-598
    _conv(spec, prefix + ".double_conv.0", out, inp, 3); _bn(spec, prefix + ".double_conv.1", out)
    _conv(spec, prefix + ".double_conv.3", out, out, 3); _bn(spec, prefix + ".double_conv.4", out)
    _conv(spec, prefix + ".identity.0", out, inp, 1); _bn(spec, prefix + ".identity.1", out)


def _up(spec, prefix, in1, out, in2=0, attn=False):                # TransFuse.py:523-549
    _double_conv(spec, prefix + ".conv", in1 + in2, out)
    if attn:                                                        # Attention_block(F_g=in1, F_l=in2, F_int=out)  :552-576
        _conv(spec, prefix + ".attn_block.W_g.0", out, in1, 1); _bn(spec, prefix + ".attn_block.W_g.1", out)
        _conv(spec, prefix + ".attn_block.W_x.0", out, in2, 1); _bn(spec, prefix + ".attn_block.W_x.1", out)
        _conv(spec, prefix + ".attn_block.psi.0", 1, out, 1); _bn(spec, prefix + ".attn_block.psi.1", 1)


def _bifusion(spec, prefix, ch1, ch2, r2, ch_int, ch_out):          # TransFuse.py:25-76
    _conv(spec, prefix + ".fc1", ch2 // r2, ch2, 1); _conv(spec, prefix + ".fc2", ch2, ch2 // r2, 1)
    _Conv(spec, prefix + ".spatial", 2, 1, 7, bn=True, bias=False)
    _Conv(spec, prefix + ".W_g", ch1, ch_int, 1, bn=True); _Conv(spec, prefix + ".W_x", ch2, ch_int, 1, bn=True)
    _Conv(spec, prefix + ".W", ch_int, ch_int, 3, bn=True)
    inp = ch1 + ch2 + ch_int; half = ch_out // 2                   # Residual(inp, ch_out)  :601-636
    _bn(spec, prefix + ".residual.bn1", inp); _Conv(spec, prefix + ".residual.conv1", inp, half, 1)
    _bn(spec, prefix + ".residual.bn2", half); _Conv(spec, prefix + ".residual.conv2", half, half, 3)
    _bn(spec, prefix + ".residual.bn3", half); _Conv(spec, prefix + ".residual.conv3", half, ch_out, 1)
    _Conv(spec, prefix + ".residual.skip_layer", inp, ch_out, 1)


def param_spec(num_domains: int = NUM_DOMAINS) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    # resnet34 without layer4 / fc
    _conv(s, "resnet.conv1", 64, 3, 7, bias=False); _bn(s, "resnet.bn1", 64)
    inp = 64
    for li, (c, n) in enumerate(((64, 3), (128, 4), (256, 6)), start=1):
        for b in range(n):
            p = f"resnet.layer{li}.{b}"
            _conv(s, p + ".conv1", c, inp if b == 0 else c, 3, bias=False); _bn(s, p + ".bn1", c)
            _conv(s, p + ".conv2", c, c, 3, bias=False); _bn(s, p + ".bn2", c)
            if b == 0 and (inp != c):
                _conv(s, p + ".downsample.0", c, inp, 1, bias=False); _bn(s, p + ".downsample.1", c)
        inp = c
    # DeiT-small-adapt (DeiT.py:116-139): cls_token is registered but unused by DeiT_adapt.forward
    s["transformer.cls_token"] = ("pos", (1, 1, EMBED)); s["transformer.pos_embed"] = ("pos", (1, 256, EMBED))
    _conv(s, "transformer.patch_embed.proj", EMBED, 3, PATCH)
    hid = max(EMBED // 2, 4)
    for i in range(DEPTH):
        b = f"transformer.blocks.{i}"
        s[b + ".norm1.weight"] = ("ln_w", (EMBED,)); s[b + ".norm1.bias"] = ("ln_b", (EMBED,))
        s[b + ".attn.qkv.weight"] = ("linear", (3 * EMBED, EMBED)); s[b + ".attn.qkv.bias"] = ("bias", (3 * EMBED,))
        s[b + ".attn.proj.weight"] = ("linear", (EMBED, EMBED)); s[b + ".attn.proj.bias"] = ("bias", (EMBED,))
        s[b + ".attn.domain_layer.0.weight"] = ("da", (hid, num_domains)); s[b + ".attn.domain_layer.0.bias"] = ("bias", (hid,))
        s[b + ".attn.domain_layer.2.weight"] = ("da", (EMBED, hid)); s[b + ".attn.domain_layer.2.bias"] = ("bias", (EMBED,))
        s[b + ".norm2.weight"] = ("ln_w", (EMBED,)); s[b + ".norm2.bias"] = ("ln_b", (EMBED,))
        s[b + ".mlp.fc1.weight"] = ("linear", (MLP_RATIO * EMBED, EMBED)); s[b + ".mlp.fc1.bias"] = ("bias", (MLP_RATIO * EMBED,))
        s[b + ".mlp.fc2.weight"] = ("linear", (EMBED, MLP_RATIO * EMBED)); s[b + ".mlp.fc2.bias"] = ("bias", (EMBED,))
    s["transformer.norm.weight"] = ("ln_w", (EMBED,)); s["transformer.norm.bias"] = ("ln_b", (EMBED,))
    _up(s, "up1", 384, 128); _up(s, "up2", 128, 64)
    _Conv(s, "final_x.0", 256, 64, 1, bn=True); _Conv(s, "final_x.1", 64, 64, 3, bn=True); _Conv(s, "final_x.2", 64, 1, 3)
    _Conv(s, "final_1.0", 64, 64, 3, bn=True); _Conv(s, "final_1.1", 64, 1, 3)
    _Conv(s, "final_2.0", 64, 64, 3, bn=True); _Conv(s, "final_2.1", 64, 1, 3)
    _bifusion(s, "up_c", 256, 384, 4, 256, 256)
    _bifusion(s, "up_c_1_1", 128, 128, 2, 128, 128); _up(s, "up_c_1_2", 256, 128, 128, attn=True)
    _bifusion(s, "up_c_2_1", 64, 64, 1, 64, 64); _up(s, "up_c_2_2", 128, 64, 64, attn=True)
    return s


def make_params(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic test weights with O(1) activations (the generator of oracle/params.py)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (kind, shape) in param_spec().items():
        n = int(np.prod(shape)) if len(shape) else 1
        u = uniform_pm1(seed, _stream_id(name), n)
        if kind == "conv":
            v = u * math.sqrt(3.0 / int(np.prod(shape[1:])))
        elif kind == "linear":
            v = u * math.sqrt(3.0 / shape[1])
        elif kind == "da":
            v = u * (1.5 if shape[1] <= 8 else 3.0 / math.sqrt(shape[1]))
        elif kind == "bias":
            v = u * 0.1
        elif kind == "pos":
            v = u * 0.2
        elif kind in ("ln_w", "bn_w", "bn_rv"):
            v = 1.0 + 0.5 * u
        elif kind in ("ln_b", "bn_b", "bn_rm"):
            v = 0.1 * u
        elif kind == "bn_nbt":
            out[name] = np.zeros((), dtype=np.int64)
            continue
        else:
            raise KeyError(kind)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def to_torch(params_np) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.asarray(v).copy()) for k, v in params_np.items()}


# ---------------------------------------------------------------------------------------------------------------------
# functional forward
# ---------------------------------------------------------------------------------------------------------------------
class TFState:
    def __init__(self, training: bool = True, update_bn: bool = True):
        self.training, self.update_bn = training, update_bn


def _bnf(P, prefix, x, st: TFState):
    rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    if st.training and not st.update_bn:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, P[prefix + ".weight"], P[prefix + ".bias"], st.training, 0.1, 1e-5)


def _convf(P, prefix, x, stride=1):
    w = P[prefix + ".weight"]
    return F.conv2d(x, w, P.get(prefix + ".bias"), stride, (w.shape[-1] - 1) // 2)


def _Convf(P, prefix, x, st, bn, relu):                             # TransFuse.py:651-656
    x = _convf(P, prefix + ".conv", x)
    if bn:
        x = _bnf(P, prefix + ".bn", x, st)
    return F.relu(x) if relu else x


def _double_convf(P, prefix, x, st):                                # TransFuse.py:597-598
    a = F.relu(_bnf(P, prefix + ".double_conv.1", _convf(P, prefix + ".double_conv.0", x), st))
    a = _bnf(P, prefix + ".double_conv.4", _convf(P, prefix + ".double_conv.3", a), st)
    b = _bnf(P, prefix + ".identity.1", _convf(P, prefix + ".identity.0", x), st)
    return F.relu(a + b)


def _upf(P, prefix, x1, st, x2=None):                               # TransFuse.py:535-549
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    if x2 is not None:
        assert x1.shape[2:] == x2.shape[2:]                         # diffX = diffY = 0 at 256x256: F.pad is the identity
        if (prefix + ".attn_block.psi.0.weight") in P:              # Attention_block(g = x1, x = x2)  :570-576
            g1 = _bnf(P, prefix + ".attn_block.W_g.1", _convf(P, prefix + ".attn_block.W_g.0", x1), st)
            xx = _bnf(P, prefix + ".attn_block.W_x.1", _convf(P, prefix + ".attn_block.W_x.0", x2), st)
            psi = F.relu(g1 + xx)
            psi = torch.sigmoid(_bnf(P, prefix + ".attn_block.psi.1", _convf(P, prefix + ".attn_block.psi.0", psi), st))
            x2 = x2 * psi
        x1 = torch.cat([x2, x1], dim=1)
    return _double_convf(P, prefix + ".conv", x1, st)


def _residualf(P, prefix, x, st):                                   # TransFuse.py:619-636 (need_skip: inp_dim != out_dim in all three uses)
    residual = _convf(P, prefix + ".skip_layer.conv", x)
    out = F.relu(_bnf(P, prefix + ".bn1", x, st))
    out = _convf(P, prefix + ".conv1.conv", out)
    out = F.relu(_bnf(P, prefix + ".bn2", out, st))
    out = _convf(P, prefix + ".conv2.conv", out)
    out = F.relu(_bnf(P, prefix + ".bn3", out, st))
    out = _convf(P, prefix + ".conv3.conv", out)
    return out + residual


def _bifusionf(P, prefix, g, x, st):                                # TransFuse.py:53-76 (drop_rate = 0 in parity runs)
    W_g = _Convf(P, prefix + ".W_g", g, st, True, False)
    W_x = _Convf(P, prefix + ".W_x", x, st, True, False)
    bp = _Convf(P, prefix + ".W", W_g * W_x, st, True, True)
    g_in = g
    gp = torch.cat((torch.max(g, 1)[0].unsqueeze(1), torch.mean(g, 1).unsqueeze(1)), dim=1)       # ChannelPool :20-22
    gs = _Convf(P, prefix + ".spatial", gp, st, True, False)
    g = torch.sigmoid(gs) * g_in
    x_in = x
    xm = x.mean((2, 3), keepdim=True)
    xm = F.relu(_convf(P, prefix + ".fc1", xm))
    xm = _convf(P, prefix + ".fc2", xm)
    x = torch.sigmoid(xm) * x_in
    return _residualf(P, prefix + ".residual", torch.cat([g, x, bp], 1), st)


def _basic_block(P, prefix, x, stride, st):
    idt = x
    if (prefix + ".downsample.0.weight") in P:
        idt = _bnf(P, prefix + ".downsample.1", F.conv2d(x, P[prefix + ".downsample.0.weight"], None, stride), st)
    y = F.relu(_bnf(P, prefix + ".bn1", F.conv2d(x, P[prefix + ".conv1.weight"], None, stride, 1), st))
    y = _bnf(P, prefix + ".bn2", F.conv2d(y, P[prefix + ".conv2.weight"], None, 1, 1), st)
    return F.relu(y + idt)


def _layer(P, prefix, x, n, stride, st):
    for b in range(n):
        x = _basic_block(P, f"{prefix}.{b}", x, stride if b == 0 else 1, st)
    return x


def attention_sup(P, prefix, x, domain_label):                      # vision_transformer.py:148-169
    B, N, C = x.shape
    qkv = F.linear(x, P[prefix + ".qkv.weight"], P[prefix + ".qkv.bias"]).reshape(B, N, 3, HEADS, C // HEADS).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (C // HEADS) ** -0.5
    attn = attn.softmax(dim=-1)
    o = attn @ v                                                     # (B,H,N,K)
    da = F.linear(F.relu(F.linear(domain_label, P[prefix + ".domain_layer.0.weight"], P[prefix + ".domain_layer.0.bias"])),
                  P[prefix + ".domain_layer.2.weight"], P[prefix + ".domain_layer.2.bias"])          # (B, H*K)
    da = torch.softmax(da.view(B, HEADS, 1, C // HEADS), dim=1)      # 'b (h k) c -> b h c k', softmax over heads
    o = (da * o).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, P[prefix + ".proj.weight"], P[prefix + ".proj.bias"])


def deit_adapt(P, imgs, domain_label):                              # DeiT.py:57-71 (drop rates 0)
    x = F.conv2d(imgs, P["transformer.patch_embed.proj.weight"], P["transformer.patch_embed.proj.bias"], PATCH).flatten(2).transpose(1, 2)
    x = x + P["transformer.pos_embed"]
    for i in range(DEPTH):
        b = f"transformer.blocks.{i}"
        n1 = F.layer_norm(x, (EMBED,), P[b + ".norm1.weight"], P[b + ".norm1.bias"], 1e-6)
        x = x + attention_sup(P, b + ".attn", n1, domain_label)
        n2 = F.layer_norm(x, (EMBED,), P[b + ".norm2.weight"], P[b + ".norm2.bias"], 1e-6)
        h = F.gelu(F.linear(n2, P[b + ".mlp.fc1.weight"], P[b + ".mlp.fc1.bias"]))
        x = x + F.linear(h, P[b + ".mlp.fc2.weight"], P[b + ".mlp.fc2.bias"])
    return F.layer_norm(x, (EMBED,), P["transformer.norm.weight"], P["transformer.norm.bias"], 1e-6)


def transfuse_forward(P, imgs: Tensor, domain_label: Tensor, st: Optional[TFState] = None):
    """TransFuse_S_adapt.forward (TransFuse.py:228-270) with drop_rate = 0 -> (map_x, map_1, map_2), logits (B,1,H,W)."""
    st = st or TFState()
    B = imgs.shape[0]
    x_b = deit_adapt(P, imgs, domain_label).transpose(1, 2).reshape(B, -1, 16, 16)
    x_b_1 = _upf(P, "up1", x_b, st)
    x_b_2 = _upf(P, "up2", x_b_1, st)
    x_u = F.relu(_bnf(P, "resnet.bn1", F.conv2d(imgs, P["resnet.conv1.weight"], None, 2, 3), st))
    x_u = F.max_pool2d(x_u, 3, 2, 1)
    x_u_2 = _layer(P, "resnet.layer1", x_u, 3, 1, st)
    x_u_1 = _layer(P, "resnet.layer2", x_u_2, 4, 2, st)
    x_u = _layer(P, "resnet.layer3", x_u_1, 6, 2, st)
    x_c = _bifusionf(P, "up_c", x_u, x_b, st)
    x_c_1_1 = _bifusionf(P, "up_c_1_1", x_u_1, x_b_1, st)
    x_c_1 = _upf(P, "up_c_1_2", x_c, st, x_c_1_1)
    x_c_2_1 = _bifusionf(P, "up_c_2_1", x_u_2, x_b_2, st)
    x_c_2 = _upf(P, "up_c_2_2", x_c_1, st, x_c_2_1)

    def head(prefix, x, convs):
        for i, (bn, relu) in enumerate(convs):
            x = _Convf(P, f"{prefix}.{i}", x, st, bn, relu)
        return x
    map_x = F.interpolate(head("final_x", x_c, [(True, True), (True, True), (False, False)]), scale_factor=16, mode="bilinear", align_corners=True)
    map_1 = F.interpolate(head("final_1", x_b_2, [(True, True), (False, False)]), scale_factor=4, mode="bilinear", align_corners=True)
    map_2 = F.interpolate(head("final_2", x_c_2, [(True, True), (False, False)]), scale_factor=4, mode="bilinear", align_corners=True)
    return map_x, map_1, map_2


def structure_loss(pred: Tensor, mask: Tensor) -> Tensor:          # multi_train_TransFuse.py:29-38
    weit = 1 + 5 * torch.abs(F.avg_pool2d(mask, kernel_size=31, stride=1, padding=15) - mask)
    wbce = F.binary_cross_entropy_with_logits(pred, mask, reduction="none")
    wbce = (weit * wbce).sum(dim=(2, 3)) / weit.sum(dim=(2, 3))
    pred = torch.sigmoid(pred)
    inter = ((pred * mask) * weit).sum(dim=(2, 3))
    union = ((pred + mask) * weit).sum(dim=(2, 3))
    wiou = 1 - (inter + 1) / (union - inter + 1)
    return (wbce + wiou).mean()


def transfuse_train_step(P, batches, st: Optional[TFState] = None, timing=None):
    """multi_train_TransFuse.py:141-189: per domain loss = 0.5 SL(map_2) + 0.3 SL(map_1) + 0.2 SL(map_x); one backward of the sum.
    batches: [(img, label, set_id:int)].  -> (per-domain losses, {name: grad})"""
    import time
    t0 = time.perf_counter()
    st = st or TFState()
    leaves = {k: v for k, v in P.items() if v.is_floating_point() and "running_" not in k}
    for v in leaves.values():
        v.requires_grad_(True); v.grad = None
    losses, tot = [], 0.0
    for img, label, sid in batches:
        dl = F.one_hot(torch.full((img.shape[0],), sid, dtype=torch.long), NUM_DOMAINS).to(img.dtype)
        m4, m3, m2 = transfuse_forward(P, img, dl, st)
        loss = 0.5 * structure_loss(m2, label) + 0.3 * structure_loss(m3, label) + 0.2 * structure_loss(m4, label)
        losses.append(float(loss.detach())); tot = tot + loss
    t1 = time.perf_counter()
    tot.backward()
    if timing is not None:
        timing["fwd_ms"] = (t1 - t0) * 1e3; timing["bwd_ms"] = (time.perf_counter() - t1) * 1e3
    grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in leaves.items()}
    for v in leaves.values():
        v.requires_grad_(False)
    return losses, grads
