"""CPU restatement (plain torch, fp32 or fp64) of the MDViT / BASE forward path and step losses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style over a flat
``{state_dict name: tensor}`` mapping; tokens are (B, N, C) with n = h*W + w, i.e. NHWC.
Written from the math in SURVEY.md Appendix A; every function cites the reference lines it
restates (paths under /root/reference).  Pinned by tests/golden/*.npz, which
oracle/gen_golden.py produced by running the reference itself in the build container.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .params import CRPE_WINDOW

Tensor = torch.Tensor
Params = Dict[str, Tensor]


class RefState:
    """Mode flags for one oracle call (the nn.Module train()/eval() state + drop rates)."""

    def __init__(self, training: bool = True, drop_rate: float = 0.0, drop_path_rate: float = 0.0,
                 aux_drop: float = 0.0, update_bn: bool = True, adapt_method="Sup"):
        self.training = training
        self.drop_rate = drop_rate
        self.drop_path_rate = drop_path_rate
        self.aux_drop = aux_drop          # MLPDecoderFM's Dropout2d(0.1): Decoders.py:294,309
        self.update_bn = update_bn
        self.adapt_method = adapt_method
        self.kink_margin = float("inf")   # min distance of any BN output to a kink of its activation (ReLU: 0, Hardswish: +-3)


# ---- elementary pieces ------------------------------------------------------------------------

def hardswish(x: Tensor) -> Tensor:
    # x * relu6(x + 3) / 6          (nn.Hardswish; mpvit.py:81-124 act_layer)
    return x * torch.clamp(x + 3.0, 0.0, 6.0) / 6.0


def batch_norm(P: Params, prefix: str, x: Tensor, st: RefState, eps: float = 1e-5, momentum: float = 0.1, kinks=None) -> Tensor:
    """nn.BatchNorm2d on NCHW x.  Train: biased batch variance for normalisation, running stats
    updated with momentum 0.1 and the UNBIASED variance."""
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if st.training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if st.update_bn:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                P[prefix + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
                P[prefix + ".running_var"].mul_(1 - momentum).add_(momentum * var.detach() * (n / max(n - 1, 1)))
                P[prefix + ".num_batches_tracked"].add_(1)
    else:
        mean, var = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    out = xh * w[None, :, None, None] + b[None, :, None, None]
    if kinks:
        with torch.no_grad():
            st.kink_margin = min(st.kink_margin, min(float((out - k).abs().min()) for k in kinks))
    return out


def dropout(x: Tensor, p: float, st: RefState) -> Tensor:
    if not st.training or p == 0.0:
        return x
    return F.dropout(x, p, True)


def drop_path(x: Tensor, p: float, st: RefState) -> Tensor:
    """timm DropPath: per-sample Bernoulli(1-p)/(1-p) (mdvit.py:338,354,359)."""
    if not st.training or p == 0.0:
        return x
    keep = 1.0 - p
    mask = torch.empty(x.shape[0], *([1] * (x.dim() - 1)), dtype=x.dtype).bernoulli_(keep)
    return x * mask / keep


def tokens_to_image(x: Tensor, H: int, W: int) -> Tensor:
    B, N, C = x.shape
    return x.transpose(1, 2).reshape(B, C, H, W)


def image_to_tokens(x: Tensor) -> Tensor:
    return x.flatten(2).transpose(1, 2)


# ---- transformer block ------------------------------------------------------------------------

def conv_pos_enc(P: Params, prefix: str, x: Tensor, H: int, W: int) -> Tensor:
    """ConvPosEnc.forward, mpvit.py:239-248:  x + dwconv3x3_bias(x) on the token image."""
    C = x.shape[2]
    img = tokens_to_image(x, H, W)
    y = F.conv2d(img, P[prefix + ".proj.weight"], P[prefix + ".proj.bias"], 1, 1, 1, C) + img
    return image_to_tokens(y)


def conv_rel_pos_enc(P: Params, prefix: str, q: Tensor, v: Tensor, H: int, W: int) -> Tensor:
    """ConvRelPosEnc.forward, mpvit.py:296-318: q * dwconv_{3|5|7}(v-as-image), channel = head*Ch+ch."""
    B, h, N, Ch = q.shape
    vimg = v.permute(0, 1, 3, 2).reshape(B, h * Ch, H, W)
    outs, c0 = [], 0
    for wi, (win, split) in enumerate(CRPE_WINDOW):
        c1 = c0 + split * Ch
        outs.append(F.conv2d(vimg[:, c0:c1], P[f"{prefix}.conv_list.{wi}.weight"], P[f"{prefix}.conv_list.{wi}.bias"],
                             1, win // 2, 1, split * Ch))
        c0 = c1
    conv_v = torch.cat(outs, 1).reshape(B, h, Ch, N).permute(0, 1, 3, 2)
    return q * conv_v


def domain_attention(P: Params, prefix: str, domain_label: Tensor, h: int) -> Tensor:
    """mdvit.py:272-276,301-304: softmax over HEADS of Linear(ReLU(Linear(one_hot))) -> (B,h,1,Ch)."""
    z = F.linear(domain_label, P[prefix + ".domain_layer.0.weight"], P[prefix + ".domain_layer.0.bias"])
    z = F.linear(torch.relu(z), P[prefix + ".domain_layer.2.weight"], P[prefix + ".domain_layer.2.bias"])
    B, C = z.shape
    return torch.softmax(z.reshape(B, h, 1, C // h), dim=1)


def factor_att(P: Params, prefix: str, crpe_prefix: str, x: Tensor, H: int, W: int, heads: int,
               domain_label: Optional[Tensor], st: RefState) -> Tensor:
    """FactorAtt_ConvRelPosEnc_Sup.forward (mdvit.py:281-313) / FactorAtt_ConvRelPosEnc.forward
    (mpvit.py:347-373) when domain_label is None."""
    B, N, C = x.shape
    Ch = C // heads
    qkv = F.linear(x, P[prefix + ".qkv.weight"], P[prefix + ".qkv.bias"]).reshape(B, N, 3, heads, Ch).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    ks = torch.softmax(k, dim=2)                      # over the TOKEN axis
    M = ks.transpose(2, 3) @ v                        # (B,h,Ch,Ch)
    fa = q @ M
    y = (Ch ** -0.5) * fa + conv_rel_pos_enc(P, crpe_prefix, q, v, H, W)
    if domain_label is not None:
        y = domain_attention(P, prefix, domain_label, heads) * y
    y = y.transpose(1, 2).reshape(B, N, C)
    y = F.linear(y, P[prefix + ".proj.weight"], P[prefix + ".proj.bias"])
    return dropout(y, st.drop_rate, st)


def mlp(P: Params, prefix: str, x: Tensor, st: RefState) -> Tensor:
    """Mlp.forward, mpvit.py:71-78 (exact-erf GELU)."""
    y = F.gelu(F.linear(x, P[prefix + ".fc1.weight"], P[prefix + ".fc1.bias"]))
    y = dropout(y, st.drop_rate, st)
    y = F.linear(y, P[prefix + ".fc2.weight"], P[prefix + ".fc2.bias"])
    return dropout(y, st.drop_rate, st)


def serial_block(P: Params, stage: str, i: int, x: Tensor, H: int, W: int, heads: int,
                 domain_label: Optional[Tensor], st: RefState) -> Tensor:
    """SerialBlock_adapt.forward, mdvit.py:346-361."""
    b = f"{stage}.mhca_blks.{i}"
    C = x.shape[2]
    x = conv_pos_enc(P, f"{stage}.cpe", x, H, W)
    cur = F.layer_norm(x, (C,), P[b + ".norm1.weight"], P[b + ".norm1.bias"], 1e-6)
    cur = factor_att(P, b + ".factoratt_crpe", f"{stage}.crpe", cur, H, W, heads, domain_label, st)
    x = x + drop_path(cur, st.drop_path_rate, st)
    cur = F.layer_norm(x, (C,), P[b + ".norm2.weight"], P[b + ".norm2.bias"], 1e-6)
    cur = mlp(P, b + ".mlp", cur, st)
    return x + drop_path(cur, st.drop_path_rate, st)


def mhsa_stage(P: Params, stage: str, x: Tensor, H: int, W: int, heads: int, layers: int,
               domain_label: Optional[Tensor], st: RefState) -> Tensor:
    """MHSA_stage_adapt.forward, mdvit.py:437-440."""
    for i in range(layers):
        x = serial_block(P, stage, i, x, H, W, heads, domain_label, st)
    return x


# ---- conv blocks ------------------------------------------------------------------------------

def conv_bn_hswish(P: Params, prefix: str, x: Tensor, stride: int, st: RefState) -> Tensor:
    """Conv2d_BN (mpvit.py:81-124) as used by the stem (mdvit.py:509-526)."""
    y = F.conv2d(x, P[prefix + ".conv.weight"], None, stride, 1)
    return hardswish(batch_norm(P, prefix + ".bn", y, st, kinks=(-3.0, 3.0)))


def dw_patch_embed(P: Params, prefix: str, x: Tensor, stride: int, st: RefState) -> Tensor:
    """DWConv2d_BN, mdvit flavour (mdvit.py:74-123): dw3x3(groups=in) -> pw1x1 -> BN -> Hardswish."""
    cin = x.shape[1]
    y = F.conv2d(x, P[prefix + ".dwconv.weight"], None, stride, 1, 1, cin)
    y = F.conv2d(y, P[prefix + ".pwconv.weight"])
    return hardswish(batch_norm(P, prefix + ".bn", y, st, kinks=(-3.0, 3.0)))


def bridge(P: Params, x: Tensor, st: RefState) -> Tensor:
    """mdvit.py:557-564."""
    y = torch.relu(batch_norm(P, "bridge.1", F.conv2d(x, P["bridge.0.weight"], P["bridge.0.bias"], 1, 1), st, kinks=(0.0,)))
    return torch.relu(batch_norm(P, "bridge.4", F.conv2d(y, P["bridge.3.weight"], P["bridge.3.bias"], 1, 1), st, kinks=(0.0,)))


def decoder_block(P: Params, j, x: Tensor, skip: Tensor, heads: int, layers: int,
                  domain_label: Optional[Tensor], st: RefState) -> Tensor:
    """UnetDecodingBlockTransformer.forward (use_res=False), Decoders.py:194-214, with the
    Decoders.py flavour of DWConv2d_BN (:15-63: Conv2d(2*out, out, 3, groups=out)).  j: 1..4 -> decoder{j}, or the
    module prefix itself (the 'Transformer' peer decoders debranchs.{d}.{j})."""
    p = f"decoder{j}" if isinstance(j, int) else j
    H, W = skip.shape[2:]
    u = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
    u = F.conv2d(u, P[p + ".conv_before.weight"], P[p + ".conv_before.bias"])
    z = torch.cat((skip, u), 1)
    cout = skip.shape[1]
    z = F.conv2d(z, P[p + ".conv_after.dwconv.weight"], None, 1, 1, 1, cout)
    z = F.conv2d(z, P[p + ".conv_after.pwconv.weight"])
    z = hardswish(batch_norm(P, p + ".conv_after.bn", z, st, kinks=(-3.0, 3.0)))
    t = mhsa_stage(P, p + ".mhsa_block", image_to_tokens(z), H, W, heads, layers, domain_label, st)
    return tokens_to_image(t, H, W)


def aux_head(P: Params, d: int, feats: Sequence[Tensor], img_size, st: RefState) -> Tensor:
    """MLPDecoderFM.forward, Decoders.py:315-339 -- or MLPDecoder.forward, Decoders.py:262-286, when the fuse conv takes the
    four encoder features only (decoder_name='MLP'); d = 1..4 picks debranch{d} (mdvit.py:715-724)."""
    p = f"debranch{d}"
    h, w = feats[0].shape[2:]
    ups = []
    for q in range(4):
        y = F.conv2d(feats[q], P[f"{p}.linear{q + 1}.weight"], P[f"{p}.linear{q + 1}.bias"])
        ups.append(F.interpolate(y, size=(h, w), mode="bilinear", align_corners=False))
    with_fm = P[p + ".linear_fuse.0.weight"].shape[1] > sum(u.shape[1] for u in ups)      # MLPFM: + the main decoder's feature
    y = torch.cat(ups + ([feats[4]] if with_fm else []), 1)
    y = F.conv2d(y, P[p + ".linear_fuse.0.weight"], P[p + ".linear_fuse.0.bias"])
    y = torch.relu(batch_norm(P, p + ".linear_fuse.1", y, st, kinks=(0.0,)))
    if st.training and st.aux_drop > 0:
        y = F.dropout2d(y, st.aux_drop, True)
    y = F.interpolate(y, size=tuple(img_size), mode="bilinear", align_corners=False)
    return F.conv2d(y, P[p + ".linear_out.weight"], P[p + ".linear_out.bias"])


def deeplab_head(P: Params, d: int, feats: Sequence[Tensor], img_size, st: RefState) -> Tensor:
    """DeepLabV3Decoder.forward, Decoders.py:229-236, over ASPP (Utils/_deeplab.py:115-166, rates 6/12/18) on encoder_outs[-1]."""
    p = f"debranch{d}.classifier"
    x = feats[3]
    a = p + ".0"
    res = [torch.relu(batch_norm(P, a + ".convs.0.1", F.conv2d(x, P[a + ".convs.0.0.weight"]), st, kinks=(0.0,)))]
    for i, r in ((1, 6), (2, 12), (3, 18)):
        res.append(torch.relu(batch_norm(P, f"{a}.convs.{i}.1", F.conv2d(x, P[f"{a}.convs.{i}.0.weight"], None, 1, r, r), st, kinks=(0.0,))))
    g = F.adaptive_avg_pool2d(x, 1)
    g = torch.relu(batch_norm(P, a + ".convs.4.2", F.conv2d(g, P[a + ".convs.4.1.weight"]), st, kinks=(0.0,)))
    res.append(F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False))
    y = torch.relu(batch_norm(P, a + ".project.1", F.conv2d(torch.cat(res, 1), P[a + ".project.0.weight"]), st, kinks=(0.0,)))
    if st.training and st.aux_drop > 0:
        y = F.dropout(y, st.aux_drop, True)              # nn.Dropout(0.1), _deeplab.py:155
    y = torch.relu(batch_norm(P, p + ".2", F.conv2d(y, P[p + ".1.weight"], None, 1, 1), st, kinks=(0.0,)))
    y = F.conv2d(y, P[p + ".4.weight"], P[p + ".4.bias"])
    return F.interpolate(y, size=tuple(img_size), mode="bilinear", align_corners=False)


# ---- whole models -----------------------------------------------------------------------------

def _encoder_decoder(P: Params, x: Tensor, domain_label: Optional[Tensor], st: RefState,
                     heads=(8, 8, 8, 8), layers=(2, 2, 2, 2)):
    img_size = x.shape[2:]
    x = conv_bn_hswish(P, "stem.0", x, 2, st)
    x = conv_bn_hswish(P, "stem.1", x, 2, st)
    enc = []
    for s in range(4):
        x = dw_patch_embed(P, f"patch_embed_stages.{s}.patch_conv", x, 1 if s == 0 else 2, st)
        H, W = x.shape[2:]
        t = mhsa_stage(P, f"mhsa_stages.{s}", image_to_tokens(x), H, W, heads[s], layers[s], domain_label, st)
        x = tokens_to_image(t, H, W)
        enc.append(x)
    out = bridge(P, enc[3], st)
    st.bridge_out = out                       # (the 'Transformer' peer decoders start from it, mdvit.py:706-709)
    for j in range(1, 5):
        s = 4 - j
        out = decoder_block(P, j, out, enc[s], heads[s], layers[s], domain_label, st)
    dec4 = out
    up = F.interpolate(dec4, size=tuple(img_size), mode="bilinear", align_corners=False)
    logits = F.conv2d(up, P["finalconv.0.weight"], P["finalconv.0.bias"])
    return logits, enc, dec4, img_size


def mdvit_forward(P: Params, x: Tensor, domain_label: Optional[Tensor] = None, d: Optional[str] = None,
                  st: Optional[RefState] = None):
    """MDViT.forward, mdvit.py:667-730 (decoder_name='MLPFM') -> [out, aux_out]."""
    st = st or RefState()
    if st.adapt_method != "Sup":
        domain_label = None
    logits, enc, dec4, img_size = _encoder_decoder(P, x, domain_label, st)
    aux = None
    if "debranchs.0.4.0.weight" in P:          # decoder_name='Transformer' (mdvit.py:705-713): the domain's own decoder, no adapter
        k = int(d)
        a = st.bridge_out
        for j in range(4):
            s = 3 - j
            a = decoder_block(P, f"debranchs.{k}.{j}", a, enc[s], 8, 2, None, st)
        a = F.interpolate(a, size=tuple(img_size), mode="bilinear", align_corners=False)
        aux = F.conv2d(a, P[f"debranchs.{k}.4.0.weight"], P[f"debranchs.{k}.4.0.bias"])
    elif d in ("0", "1", "2", "3"):
        head = deeplab_head if f"debranch{int(d) + 1}.classifier.4.weight" in P else aux_head
        aux = head(P, int(d) + 1, enc + [dec4], img_size, st)
    return [logits, aux]


def dsn_view(P: Params, d: int, canonical_names) -> Params:
    """The MDViT-named view of an MDViT_DSN parameter dict for domain d: norms[d] selected (mdvit.py:63-70,169-178,
    396-412,911-917, Decoders.py:106-118).  Shares the tensors, so BN running statistics update in place."""
    from .params import dsn_name
    return {n: P[dsn_name(n, d)[0]] for n in canonical_names}


def mdvit_dsn_forward(P: Params, x: Tensor, domain_label: Optional[Tensor] = None, d: Optional[str] = None,
                      st: Optional[RefState] = None):
    """MDViT_DSN.forward, mdvit.py:896-960 (decoder_name='MLPFM'): MDViT.forward on the domain's norms."""
    from .params import param_spec
    names = param_spec("MDViT", "Sup").keys()
    return mdvit_forward(dsn_view(P, int(d), names), x, domain_label, d, st)


def base_dsn_forward(P: Params, x: Tensor, domain_label: Optional[Tensor] = None, d: Optional[str] = None,
                     st: Optional[RefState] = None) -> Tensor:
    """BASE_DSN.forward, base.py:651-700: BASE.forward on the norms of domain int(d)."""
    from .params import param_spec
    names = param_spec("BASE", st.adapt_method if st is not None else False).keys()
    return base_forward(dsn_view(P, int(d), names), x, domain_label, st)


def base_forward(P: Params, x: Tensor, domain_label: Optional[Tensor] = None, st: Optional[RefState] = None) -> Tensor:
    """BASE.forward, base.py:477-512 -> logits tensor."""
    st = st or RefState()
    return _encoder_decoder(P, x, domain_label, st)[0]


# ---- losses ------------------------------------------------------------------------------------

def dice_loss(score: Tensor, target: Tensor) -> Tensor:
    """Utils/losses.py:8-16."""
    target = target.to(score.dtype)
    smooth = 1e-5
    inter = torch.sum(score * target)
    return 1 - (2 * inter + smooth) / (torch.sum(score * score) + torch.sum(target * target) + smooth)


class _BCE(torch.autograd.Function):
    """nn.BCELoss semantics incl. its backward at saturation: forward clamps log at -100,
    backward is (p - y) / max(p (1-p), 1e-12) / N  (ATen binary_cross_entropy_backward)."""

    @staticmethod
    def forward(ctx, p, y):
        ctx.save_for_backward(p, y)
        lp = torch.clamp(torch.log(p), min=-100.0)
        l1p = torch.clamp(torch.log(1 - p), min=-100.0)
        return -(y * lp + (1 - y) * l1p).mean()

    @staticmethod
    def backward(ctx, g):
        p, y = ctx.saved_tensors
        return g * (p - y) / torch.clamp((1 - p) * p, min=1e-12) / p.numel(), None


def bce_loss(p: Tensor, y: Tensor) -> Tensor:
    """nn.BCELoss (mean; log clamped at -100), multi_train_MDViT.py:76."""
    return _BCE.apply(p, y.to(p.dtype))


def domain_losses(out: Tensor, aux: Tensor, label: Tensor):
    """multi_train_MDViT.py:147-169 for one domain: (loss, aux_loss, kt_loss)."""
    o, a = torch.sigmoid(out), torch.sigmoid(aux)
    return bce_loss(o, label) + dice_loss(o, label), bce_loss(a, label) + dice_loss(a, label), dice_loss(a, o)


def mdvit_train_step(P: Params, batches, st: Optional[RefState] = None, alpha: float = 0.5, forward=None, timing=None):
    """One optimisation step's losses and gradients, multi_train_MDViT.py:129-207.

    batches: list of (img, label, set_id:int) -- one per domain.  Returns (losses dict, grads dict).
    The aux sweep runs with every ``domain_layer`` parameter frozen, the uni sweep with all
    parameters live; gradients accumulate.  timing: optional dict, receives fwd_ms / bwd_ms (bench.py's CPU baseline)."""
    import time as _time
    _t0 = _time.perf_counter()
    st = st or RefState()
    leaves = {k: v for k, v in P.items() if v.is_floating_point() and "running_" not in k}
    for v in leaves.values():
        v.requires_grad_(True)
        v.grad = None
    tot, tot_aux, tot_kt = 0.0, 0.0, 0.0
    for img, label, sid in batches:
        dl = F.one_hot(torch.full((img.shape[0],), sid, dtype=torch.long), 4).to(img.dtype)
        out, aux = (forward or mdvit_forward)(P, img, dl, str(sid), st)
        l, la, lk = domain_losses(out, aux, label)
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    da = [v for k, v in leaves.items() if "domain_layer" in k]
    _t1 = _time.perf_counter()
    for v in da:
        v.requires_grad_(False)
    tot_aux.backward(retain_graph=True)
    for v in da:
        v.requires_grad_(True)
    uni = alpha * tot_kt + (1 - alpha) * tot
    uni.backward()
    if timing is not None:
        timing["fwd_ms"] = (_t1 - _t0) * 1e3
        timing["bwd_ms"] = (_time.perf_counter() - _t1) * 1e3
    grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in leaves.items()}
    losses = {"loss": float(tot.detach()), "aux_loss": float(tot_aux.detach()), "kt_loss": float(tot_kt.detach())}
    for v in leaves.values():
        v.requires_grad_(False)
    return losses, grads


def base_train_step(P: Params, img: Tensor, label: Tensor, domain_label: Optional[Tensor] = None,
                    st: Optional[RefState] = None, forward=None, timing=None):
    """multi_train_BASE.py:168-200 for one domain: loss = BCE+Dice, single backward.  forward: base_forward, or a closure over
    base_dsn_forward with the domain id."""
    st = st or RefState()
    base_forward_ = forward or base_forward
    leaves = {k: v for k, v in P.items() if v.is_floating_point() and "running_" not in k}
    for v in leaves.values():
        v.requires_grad_(True)
        v.grad = None
    import time as _time
    _t0 = _time.perf_counter()
    o = torch.sigmoid(base_forward_(P, img, domain_label, st))
    loss = bce_loss(o, label) + dice_loss(o, label)
    _t1 = _time.perf_counter()
    loss.backward()
    if timing is not None:
        timing["fwd_ms"] = (_t1 - _t0) * 1e3
        timing["bwd_ms"] = (_time.perf_counter() - _t1) * 1e3
    grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in leaves.items()}
    for v in leaves.values():
        v.requires_grad_(False)
    return float(loss.detach()), grads


def to_torch(params_np, dtype=torch.float32) -> Params:
    out = {}
    for k, v in params_np.items():
        t = torch.from_numpy(v.copy())
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def kink_margin(P: Params, batches, st: Optional[RefState] = None) -> float:
    """Smallest distance of any BatchNorm output to a kink of the activation that follows it, over one
    multi-domain forward.  ReLU' and Hardswish' jump there, so a datum closer than float round-off
    (~1e-6) makes ANY two fp32 implementations disagree on that element's gradient mask; test data are
    chosen with a margin well above that (see tests/ and oracle/gen_golden.py)."""
    st = st or RefState()
    P = {k: v.clone() for k, v in P.items()}
    with torch.no_grad():
        for img, label, sid in batches:
            dl = F.one_hot(torch.full((img.shape[0],), sid, dtype=torch.long), 4).to(img.dtype)
            mdvit_forward(P, img, dl, str(sid), st)
    return st.kink_margin
