"""Generate tests/golden/*.npz by running the REAL reference (imported read-only from
/root/reference) on build-owned deterministic weights and inputs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Run in the build container only:

    python -m oracle.gen_golden

Fixtures hold data only (expected outputs, losses, gradient norms/samples); inputs and weights
are re-derived at test time from oracle.params (splitmix64 counter generator), so nothing of the
reference's source travels.  Reference call sites exercised:
  multi_train_MDViT.py:57-60,129-207 (construction, 4-domain step, two-sweep backward),
  multi_train_BASE.py:66-68,168-200, mdvit.py:281-313 (FactorAtt_ConvRelPosEnc_Sup),
  mdvit.py:346-361 (SerialBlock_adapt), Decoders.py:194-214,315-339, Utils/losses.py:8-16.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from .params import make_params, uniform_pm1, param_spec, is_buffer
from .ref_import import import_reference, load_params_into

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---- deterministic inputs (shared with the tests through this module) ---------------------------

def synth_image(seed: int, B: int, H: int, W: int) -> torch.Tensor:
    u = uniform_pm1(seed, 0x1A6E, B * 3 * H * W).reshape(B, 3, H, W)
    return torch.from_numpy((2.0 * u).astype(np.float32))


def synth_label(seed: int, B: int, H: int, W: int) -> torch.Tensor:
    """Filled-ellipse binary masks (SURVEY.md 8d shape family)."""
    u = uniform_pm1(seed, 0x1ABE1, B * 4).reshape(B, 4) * 0.5 + 0.5
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    out = np.zeros((B, 1, H, W), np.float32)
    for b in range(B):
        cy, cx = (0.3 + 0.4 * u[b, 0]) * H, (0.3 + 0.4 * u[b, 1]) * W
        ry, rx = (0.1 + 0.25 * u[b, 2]) * H, (0.1 + 0.25 * u[b, 3]) * W
        out[b, 0] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.float32)
    return torch.from_numpy(out)


def synth_tokens(seed: int, stream: int, shape) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(uniform_pm1(seed, stream, n).astype(np.float32).reshape(shape))


def grad_digest(named_grads):
    """name-sorted arrays: L2 norm and the first 4 elements of every gradient."""
    names = sorted(named_grads)
    norms = np.array([0.0 if named_grads[n] is None else float(named_grads[n].double().norm()) for n in names])
    heads = np.zeros((len(names), 4), np.float32)
    for i, n in enumerate(names):
        g = named_grads[n]
        if g is not None:
            f = g.reshape(-1)[:4].float().numpy()
            heads[i, : f.size] = f
    return names, norms, heads


def _sample(t: torch.Tensor, stride: int = 37) -> np.ndarray:
    return t.detach().reshape(-1)[::stride].float().numpy().copy()


def _build_mdvit(ns, pn, img_size, adapt="Sup", decoder_name="MLPFM"):
    m = ns.MDViT(img_size=img_size, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d,
                 adapt_method=adapt, num_domains=4, decoder_name=decoder_name)
    load_params_into(m, pn)
    for d in range(1, 5):                 # Dropout2d(0.1) is a fixed default of MLPDecoderFM (Decoders.py:294)
        if decoder_name == "DeepLabV3":   # ... and Dropout(0.1) of the ASPP projection (Utils/_deeplab.py:155)
            getattr(m, f"debranch{d}").classifier[0].project[3].p = 0.0
        elif hasattr(m, f"debranch{d}"):
            getattr(m, f"debranch{d}").dropout.p = 0.0
    return m


def gen_mdvit_mlp_step(ns, S=64, B=2, seed=6):
    """the same step with decoder_name='MLP' peer heads (MLPDecoder, Decoders.py:239-286; mdvit.py:601-606)"""
    return gen_mdvit_step(ns, S, B, seed, decoder_name="MLP")


def gen_mdvit_transformer_step(ns, S=64, B=2, seed=7):
    """the same step with decoder_name='Transformer': per-domain transformer peer decoders (mdvit.py:614-642,705-713)"""
    return gen_mdvit_step(ns, S, B, seed, decoder_name="Transformer")


def gen_mdvit_deeplab_step(ns, S=64, B=2, seed=8):
    """the same step with decoder_name='DeepLabV3' peer heads (Decoders.py:218-236, Utils/_deeplab.py:115-166)"""
    return gen_mdvit_step(ns, S, B, seed, decoder_name="DeepLabV3")


def gen_mdvit_deeplab_step_b4(ns, S=64, B=4, seed=8):
    """the DeepLabV3-peer step with FOUR images per domain: the ASPP pooling branch (Utils/_deeplab.py:124-135 via mdvit.py:607-612) normalises the B pooled vectors
    of a domain batch with a train-mode BatchNorm; with B = 2 that backward divides by a two-sample variance and amplifies fp32 round-off of its input by ~1e5 (the
    B = 2 fixture then pins nothing about the branch's gradients: VERDICT r05).  Four samples condition it; this is the fixture the gradient bounds are held against."""
    return gen_mdvit_step(ns, S, B, seed, decoder_name="DeepLabV3")


def gen_mdvit_step(ns, S=64, B=2, seed=0, decoder_name="MLPFM"):
    """4-domain two-sweep step, train mode -- multi_train_MDViT.py:129-207."""
    pn = make_params(seed, model="MDViT", adapt_method="Sup", decoder_name=decoder_name)
    m = _build_mdvit(ns, pn, S, decoder_name=decoder_name).train()
    out = {}
    tot = tot_aux = tot_kt = 0.0
    bce = torch.nn.BCELoss()
    for d in range(4):
        img, lab = synth_image(100 + d, B, S, S), synth_label(200 + d, B, S, S)
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
        o, a = m(img, dl, str(d))
        out[f"out_{d}"], out[f"aux_{d}"] = o.detach().numpy().copy(), a.detach().numpy().copy()
        so, sa = torch.sigmoid(o), torch.sigmoid(a)
        l = bce(so, lab) + ns.dice_loss(so, lab)
        la = bce(sa, lab) + ns.dice_loss(sa, lab)
        lk = ns.dice_loss(sa, so)
        out[f"losses_{d}"] = np.array([float(l), float(la), float(lk)])
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    m.zero_grad()
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    da_none = all(p.grad is None for n, p in m.named_parameters() if "domain_layer" in n)
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = True
    (0.5 * tot_kt + 0.5 * tot).backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    names, norms, heads = grad_digest(grads)
    sd = m.state_dict()
    bn_names = sorted(k for k, (kind, _) in param_spec("MDViT", "Sup", decoder_name=decoder_name).items() if kind in ("bn_rm", "bn_rv"))
    out.update(grad_names=np.array(names), grad_norms=norms, grad_heads=heads,
               da_grad_none_after_aux_sweep=np.array(da_none), n_state_dict_keys=np.array(len(m.state_dict())),
               bn_names=np.array(bn_names), bn_sums=np.array([float(sd[k].double().sum()) for k in bn_names]),
               total_losses=np.array([float(tot), float(tot_aux), float(tot_kt)]),
               meta=np.array([S, B, seed]))
    # a few full gradients of small tensors for a stricter check
    for n in ("finalconv.0.weight", "mhsa_stages.0.mhca_blks.0.factoratt_crpe.domain_layer.0.weight",
              "mhsa_stages.0.cpe.proj.weight", "mhsa_stages.0.crpe.conv_list.2.weight",
              "decoder4.conv_after.dwconv.weight", "stem.0.conv.weight", "debranch2.linear_out.weight",
              "debranchs.2.4.0.weight", "debranchs.1.3.conv_after.dwconv.weight", "debranch3.classifier.4.weight",
              "debranch1.classifier.0.convs.4.2.weight", "debranch4.classifier.0.convs.2.1.weight", "debranch2.classifier.0.project.1.bias", "debranchs.3.0.mhsa_block.mhca_blks.1.norm1.weight",
              "mhsa_stages.3.mhca_blks.1.norm2.weight", "stem.1.bn.weight"):
        if n in grads:
            out["grad::" + n] = grads[n].numpy().copy()
    return out


def gen_mdvit_dsn_step(ns, S=64, B=2, seed=4):
    """MDViT_DSN (domain-specific norms, mdvit.py:735-960): 4-domain two-sweep step, train mode."""
    pn = make_params(seed, model="MDViT_DSN", adapt_method="Sup")
    m = ns.MDViT_DSN(img_size=S, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                     num_domains=4, decoder_name="MLPFM")
    load_params_into(m, pn)
    for d in range(1, 5):
        getattr(m, f"debranch{d}").dropout.p = 0.0
    m.train()
    out = {}
    tot = tot_aux = tot_kt = 0.0
    bce = torch.nn.BCELoss()
    for d in range(4):
        img, lab = synth_image(500 + d, B, S, S), synth_label(600 + d, B, S, S)
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
        o, a = m(img, dl, str(d))
        out[f"out_{d}"], out[f"aux_{d}"] = _sample(o, 7), _sample(a, 7)
        so, sa = torch.sigmoid(o), torch.sigmoid(a)
        l = bce(so, lab) + ns.dice_loss(so, lab)
        la = bce(sa, lab) + ns.dice_loss(sa, lab)
        lk = ns.dice_loss(sa, so)
        out[f"losses_{d}"] = np.array([float(l), float(la), float(lk)])
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    m.zero_grad()
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = True
    (0.5 * tot_kt + 0.5 * tot).backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    names, norms, heads = grad_digest(grads)
    sd = m.state_dict()
    bn_names = sorted(k for k, (kind, _) in param_spec("MDViT_DSN", "Sup").items() if kind in ("bn_rm", "bn_rv"))
    out.update(grad_names=np.array(names), grad_norms=norms, grad_heads=heads,
               bn_names=np.array(bn_names), bn_sums=np.array([float(sd[k].double().sum()) for k in bn_names]),
               total_losses=np.array([float(tot), float(tot_aux), float(tot_kt)]),
               n_state_dict_keys=np.array(len(sd)), meta=np.array([S, B, seed]))
    for n in ("stem_1.bns.2.weight", "mhsa_stages.1.mhca_blks.0.norm1s.3.weight", "bridge_norms2.1.bias",
              "decoder3.conv_after.bns.0.weight", "stem_2.conv.weight", "decoder2.conv_before.bias"):
        out["grad::" + n] = grads[n].numpy().copy()
    return out


def gen_mdvit_eval(ns, S=64, B=2, seed=1):
    """eval-mode forward with non-trivial running stats (mdvit.py:667-730)."""
    pn = make_params(seed, model="MDViT", adapt_method="Sup")
    m = _build_mdvit(ns, pn, S).eval()
    out = {"meta": np.array([S, B, seed])}
    with torch.no_grad():
        for d in (0, 3):
            img = synth_image(300 + d, B, S, S)
            dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
            o, a = m(img, dl, str(d))
            out[f"out_{d}"], out[f"aux_{d}"] = o.numpy().copy(), a.numpy().copy()
    return out


def gen_mdvit_fwd_rect(ns, seed=2):
    """train-mode forward at a non-square, non-64 size (96x128, B=1) -- resolution-agnostic path."""
    pn = make_params(seed, model="MDViT", adapt_method="Sup")
    m = _build_mdvit(ns, pn, 128).train()
    img = synth_image(400, 1, 96, 128)
    dl = F.one_hot(torch.tensor([1]), 4).float()
    with torch.no_grad():
        o, a = m(img, dl, "1")
    return {"out": o.numpy().copy(), "aux": a.numpy().copy(), "meta": np.array([96, 128, 1, seed])}


def gen_base_step(ns, S=64, B=2, seed=3):
    """config 1: BASE(adapt_method=False), model(img), BCE+Dice, single backward
    (multi_train_BASE.py:66-68,168-200)."""
    pn = make_params(seed, model="BASE", adapt_method=False)
    m = ns.BASE(drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method=False)
    load_params_into(m, pn)
    m.train()
    img, lab = synth_image(500, B, S, S), synth_label(600, B, S, S)
    o = m(img)
    so = torch.sigmoid(o)
    loss = torch.nn.BCELoss()(so, lab) + ns.dice_loss(so, lab)
    m.zero_grad()
    loss.backward()
    names, norms, heads = grad_digest({n: p.grad for n, p in m.named_parameters()})
    return {"out": o.detach().numpy().copy(), "loss": np.array(float(loss)), "grad_names": np.array(names),
            "grad_norms": norms, "grad_heads": heads, "meta": np.array([S, B, seed])}


def gen_base_dsn_step(ns, S=64, B=2, seed=9):
    """BASE_DSN (base.py:515-700) with adapt_method='Sup': two domains, forward(img, domain_label, d), BCE+Dice, one backward"""
    pn = make_params(seed, model="BASE_DSN", adapt_method="Sup")
    m = ns.BASE_DSN(drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4)
    load_params_into(m, pn)
    m.train()
    out = {"n_state_dict_keys": np.array(len(m.state_dict())), "meta": np.array([S, B, seed])}
    loss = 0.0
    for d in (2, 0):
        img, lab = synth_image(700 + d, B, S, S), synth_label(800 + d, B, S, S)
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
        o = m(img, dl, str(d))
        so = torch.sigmoid(o)
        loss = loss + torch.nn.BCELoss()(so, lab) + ns.dice_loss(so, lab)
        out[f"out_{d}"] = o.detach().numpy().copy()
    m.zero_grad()
    loss.backward()
    names, norms, heads = grad_digest({n: p.grad for n, p in m.named_parameters()})
    sd = m.state_dict()
    bn_names = sorted(k for k, (kind, _) in param_spec("BASE_DSN", "Sup").items() if kind in ("bn_rm", "bn_rv"))
    out.update(loss=np.array(float(loss)), grad_names=np.array(names), grad_norms=norms, grad_heads=heads,
               bn_names=np.array(bn_names), bn_sums=np.array([float(sd[k].double().sum()) for k in bn_names]))
    return out


def gen_factoratt(ns, seed=4):
    """FactorAtt_ConvRelPosEnc_Sup fwd + grads at small shapes (mdvit.py:243-313), and the plain
    variant (mpvit.py:321-373)."""
    out = {}
    for tag, (B, H, W, C) in {"c64": (2, 8, 8, 64), "c128": (2, 6, 10, 128), "c320": (1, 4, 4, 320)}.items():
        Ch = C // 8
        crpe = ns.ConvRelPosEnc(Ch=Ch, h=8, window={3: 2, 5: 3, 7: 3})
        att = ns.FactorAtt_Sup(H * W, C, num_heads=8, qkv_bias=True, shared_crpe=crpe, num_domains=4)
        k = 0
        for n, p in sorted(att.named_parameters()):
            k += 1
            scale = 1.5 if "domain_layer" in n else (0.1 if n.endswith("bias") else (3.0 / p.shape[1]) ** 0.5 if p.dim() == 2 else 0.3)
            with torch.no_grad():
                p.copy_(synth_tokens(seed, 1000 + k, tuple(p.shape)) * scale)
        x = synth_tokens(seed, 1, (B, H * W, C)).requires_grad_(True)
        dom = torch.tensor([1, 3][:B])
        dl = F.one_hot(dom, 4).float()
        y = att(x, (H, W), dl)
        g = synth_tokens(seed, 2, tuple(y.shape))
        (y * g).sum().backward()
        out[f"{tag}_shape"] = np.array([B, H, W, C])
        out[f"{tag}_y"] = y.detach().numpy().copy()
        out[f"{tag}_dx"] = x.grad.numpy().copy()
        for n, p in sorted(att.named_parameters()):     # big weight grads: strided samples + norm only
            if p.numel() <= 16384:
                out[f"{tag}_grad::{n}"] = p.grad.numpy().copy()
            else:
                out[f"{tag}_gradsample::{n}"] = _sample(p.grad, 29)
                out[f"{tag}_gradnorm::{n}"] = np.array(float(p.grad.double().norm()))
        # DA first layer on a one-hot == column gather, bit-exact
        with torch.no_grad():
            lin = att.domain_layer[0]
            exact = all(torch.equal(lin(F.one_hot(torch.tensor([d]), 4).float())[0], lin.weight[:, d] + lin.bias) for d in range(4))
        out[f"{tag}_da_gather_bitexact"] = np.array(exact)
    return out


def gen_losses(ns, seed=5):
    """BCE / Dice / KT on sigmoid outputs incl. saturated logits (log clamp at -100)."""
    o = synth_tokens(seed, 1, (2, 1, 32, 32)) * 6.0
    a = synth_tokens(seed, 2, (2, 1, 32, 32)) * 6.0
    o.view(-1)[:8] = torch.tensor([200.0, -200.0, 120.0, -120.0, 90.0, -90.0, 40.0, -40.0])
    lab = synth_label(seed, 2, 32, 32)
    o.requires_grad_(True); a.requires_grad_(True)
    so, sa = torch.sigmoid(o), torch.sigmoid(a)
    bce = torch.nn.BCELoss()
    l = bce(so, lab) + ns.dice_loss(so, lab)
    la = bce(sa, lab) + ns.dice_loss(sa, lab)
    lk = ns.dice_loss(sa, so)
    la.backward(retain_graph=True)
    ga_aux = a.grad.clone(); a.grad = None
    (0.5 * lk + 0.5 * l).backward()
    return {"losses": np.array([float(l), float(la), float(lk)]), "d_aux_from_auxloss": ga_aux.numpy().copy(),
            "d_out_from_uni": o.grad.numpy().copy(), "d_aux_from_uni": a.grad.numpy().copy()}


def gen_transfuse_step(ns_unused=None, S=256, B=2, seed=12):
    """BASELINE configs[4]: the reference's TransFuse_S_adapt (drop_rate = 0, train mode) on two domains x B images at its only legal size
    256x256 (TransFuse.py:228-270), step loss 0.5 SL(map_2) + 0.3 SL(map_1) + 0.2 SL(map_x) with structure_loss
    (multi_train_TransFuse.py:29-38,162-172), ONE backward of the sum (:186-189)."""
    from .ref_import import import_transfuse, lift_function
    from . import transfuse_ref as T
    tf = import_transfuse()
    structure_loss = lift_function("multi_train_TransFuse.py", "structure_loss", {"torch": torch, "F": F})      # the reference's own function (:29-38)
    pn = T.make_params(seed)
    m = tf.TransFuse_S_adapt(num_classes=1, drop_rate=0.0, normal_init=False, pretrained=False, num_domains=4)
    sd = m.state_dict()
    assert set(sd) == set(pn), (sorted(set(sd) - set(pn))[:5], sorted(set(pn) - set(sd))[:5])
    load_params_into(m, pn)
    m.train()
    out = {"n_state_dict_keys": np.array(len(sd)), "meta": np.array([S, B, seed])}
    tot = 0.0
    for di, d in enumerate((1, 3)):
        img, lab = synth_image(1200 + d, B, S, S), synth_label(1300 + d, B, S, S)
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
        m4, m3, m2 = m(img, dl)
        l = 0.5 * structure_loss(m2, lab) + 0.3 * structure_loss(m3, lab) + 0.2 * structure_loss(m4, lab)
        out[f"loss_{d}"] = np.array(float(l))
        for nm, t in (("map_x", m4), ("map_1", m3), ("map_2", m2)):
            out[f"{nm}_{d}"] = _sample(t, 61)
            out[f"{nm}_{d}_sum"] = np.array([float(t.double().sum()), float(t.double().abs().sum())])
        tot = tot + l
    tot.backward()
    names, norms, heads = grad_digest({n: p.grad for n, p in m.named_parameters()})
    out["grad_names"] = np.array(names); out["grad_norms"] = norms; out["grad_heads"] = heads
    for n in ("resnet.conv1.weight", "resnet.layer3.5.conv2.weight", "transformer.blocks.0.attn.domain_layer.2.weight", "transformer.blocks.7.attn.qkv.weight",
              "transformer.pos_embed", "up_c.spatial.conv.weight", "up_c_1_2.attn_block.psi.0.weight", "final_x.2.conv.weight", "up_c.fc1.weight"):
        out["grad__" + n] = dict(m.named_parameters())[n].grad.reshape(-1)[::7].numpy().copy()
    for k in ("resnet.bn1.running_mean", "up_c.residual.bn1.running_var", "up_c_2_2.attn_block.psi.1.running_mean"):
        out["buf__" + k] = m.state_dict()[k].numpy().copy()
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ns = import_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    jobs = {"transfuse_step_256": gen_transfuse_step, "base_dsn_step_64": gen_base_dsn_step, "mdvit_deeplab_step_64": gen_mdvit_deeplab_step, "mdvit_deeplab_step_64_b4": gen_mdvit_deeplab_step_b4, "mdvit_transformer_step_64": gen_mdvit_transformer_step, "mdvit_mlp_step_64": gen_mdvit_mlp_step, "mdvit_dsn_step_64": gen_mdvit_dsn_step, "mdvit_step_64": gen_mdvit_step, "mdvit_eval_64": gen_mdvit_eval, "mdvit_fwd_96x128": gen_mdvit_fwd_rect,
            "base_step_64": gen_base_step, "factoratt_small": gen_factoratt, "losses_small": gen_losses}
    only = set(sys.argv[1:])
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        data = fn(ns)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(data)} arrays)")


if __name__ == "__main__":
    main()
