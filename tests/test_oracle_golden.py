"""The CPU oracle (oracle/mdvit_ref.py) against the fixtures the REAL reference produced
(tests/golden/*.npz, made by oracle/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mdvit_ref as R
from oracle.gen_golden import synth_image, synth_label, synth_tokens, grad_digest
from oracle.params import make_params, param_spec, alias_map

RTOL = 2e-4   # fp32 CPU restatement vs fp32 CPU reference; different op order only


def close(a, b, rtol=RTOL, name=""):
    if isinstance(a, torch.Tensor):
        a = a.detach().numpy()
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, f"{name}: rel-to-max error {err:.3e} > {rtol}"


def test_param_inventory_counts():
    spec = param_spec("MDViT", "Sup")
    n_param = sum(1 for k, (kind, _) in spec.items() if not kind.startswith("bn_r") and kind != "bn_nbt")
    n_elem = sum(int(np.prod(s)) for k, (kind, s) in spec.items() if not kind.startswith("bn_r") and kind != "bn_nbt")
    assert n_param == 432 and len(spec) == 480 and len(alias_map()) == 128      # 608 keys
    assert n_elem == 34_970_277
    base = param_spec("BASE", False)
    n_base = sum(int(np.prod(s)) for k, (kind, s) in base.items() if not kind.startswith("bn_r") and kind != "bn_nbt")
    assert abs(n_base - 27.75e6) < 0.01e6


def test_generator_is_stable():
    p = make_params(0, model="BASE", adapt_method=False)
    w = p["stem.0.conv.weight"].reshape(-1)
    # frozen values: the fixtures were produced with exactly this generator
    assert np.allclose(w[:3], make_params(0, model="BASE", adapt_method=False)["stem.0.conv.weight"].reshape(-1)[:3])
    assert abs(float(p["finalconv.0.weight"].std()) - 0.125) < 0.03


@pytest.mark.parametrize("fixture,decoder_name", [("mdvit_step_64", "MLPFM"), ("mdvit_mlp_step_64", "MLP"),
                                                  ("mdvit_transformer_step_64", "Transformer"), ("mdvit_deeplab_step_64", "DeepLabV3"),
                                                  ("mdvit_deeplab_step_64_b4", "DeepLabV3")])
def test_mdvit_two_sweep_step(golden, fixture, decoder_name):
    """decoder_name='MLP': the peer heads without the main decoder's feature (MLPDecoder, Decoders.py:239-286);
    'Transformer': per-domain transformer peer decoders (mdvit.py:614-642,705-713)"""
    g = golden(fixture)
    S, B, seed = [int(v) for v in g["meta"]]
    P = R.to_torch(make_params(seed, model="MDViT", adapt_method="Sup", decoder_name=decoder_name))
    batches = [(synth_image(100 + d, B, S, S), synth_label(200 + d, B, S, S), d) for d in range(4)]
    # forward logits per domain (fresh params each so BN buffers evolve exactly as in the fixture)
    st = R.RefState(training=True)
    P2 = {k: v.clone() for k, v in P.items()}
    for d, (img, lab, sid) in enumerate(batches):
        dl = F.one_hot(torch.full((B,), sid, dtype=torch.long), 4).float()
        with torch.no_grad():
            o, a = R.mdvit_forward(P2, img, dl, str(sid), st)
            l = R.domain_losses(o, a, lab)
        close(o, g[f"out_{d}"], name=f"out_{d}")
        close(a, g[f"aux_{d}"], name=f"aux_{d}")
        close([float(v) for v in l], g[f"losses_{d}"], name=f"losses_{d}")
    bn_names = [str(n) for n in g["bn_names"]]
    close([float(P2[k].double().sum()) for k in bn_names], g["bn_sums"], name="bn running sums")
    losses, grads = R.mdvit_train_step(P, batches, R.RefState(training=True))
    close([losses["loss"], losses["aux_loss"], losses["kt_loss"]], g["total_losses"], name="total losses")
    names, norms, heads = grad_digest(grads)
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 2e-3, f"grad norm mismatch {names[int(rel.argmax())]} {rel.max():.2e}"
    for key in g.files:
        if key.startswith("grad::"):
            close(grads[key[6:]], g[key], rtol=1e-3, name=key)
    assert bool(g["da_grad_none_after_aux_sweep"])
    if "n_state_dict_keys" in g.files:
        assert len(P) + len(alias_map("MDViT", decoder_name=decoder_name)) == int(g["n_state_dict_keys"])


def test_mdvit_eval(golden):
    g = golden("mdvit_eval_64")
    S, B, seed = [int(v) for v in g["meta"]]
    P = R.to_torch(make_params(seed, model="MDViT", adapt_method="Sup"))
    for d in (0, 3):
        img = synth_image(300 + d, B, S, S)
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
        with torch.no_grad():
            o, a = R.mdvit_forward(P, img, dl, str(d), R.RefState(training=False))
        close(o, g[f"out_{d}"], name="eval out")
        close(a, g[f"aux_{d}"], name="eval aux")


def test_mdvit_rect(golden):
    g = golden("mdvit_fwd_96x128")
    H, W, B, seed = [int(v) for v in g["meta"]]
    P = R.to_torch(make_params(seed, model="MDViT", adapt_method="Sup"))
    with torch.no_grad():
        o, a = R.mdvit_forward(P, synth_image(400, B, H, W), F.one_hot(torch.tensor([1]), 4).float(), "1",
                               R.RefState(training=True))
    close(o, g["out"], name="rect out")
    close(a, g["aux"], name="rect aux")


def test_base_step(golden):
    g = golden("base_step_64")
    S, B, seed = [int(v) for v in g["meta"]]
    P = R.to_torch(make_params(seed, model="BASE", adapt_method=False))
    img, lab = synth_image(500, B, S, S), synth_label(600, B, S, S)
    with torch.no_grad():
        o = R.base_forward({k: v.clone() for k, v in P.items()}, img, None, R.RefState(training=True, adapt_method=False))
    close(o, g["out"], name="base out")
    loss, grads = R.base_train_step(P, img, lab, None, R.RefState(training=True, adapt_method=False))
    close(loss, g["loss"], name="base loss")
    names, norms, _ = grad_digest(grads)
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 2e-3


def test_base_dsn_step(golden):
    """BASE_DSN (base.py:515-700): per-domain norm banks under BASE's forward, two domains, one backward"""
    g = golden("base_dsn_step_64")
    S, B, seed = [int(v) for v in g["meta"]]
    P = R.to_torch(make_params(seed, model="BASE_DSN", adapt_method="Sup"))
    assert len(P) + len(alias_map("BASE")) == int(g["n_state_dict_keys"])
    st = R.RefState(training=True, adapt_method="Sup")
    leaves = {k: v for k, v in P.items() if v.is_floating_point() and "running_" not in k}
    for v in leaves.values():
        v.requires_grad_(True)
    loss = 0.0
    for d in (2, 0):
        img, lab = synth_image(700 + d, B, S, S), synth_label(800 + d, B, S, S)
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
        o = R.base_dsn_forward(P, img, dl, str(d), st)
        close(o.detach(), g[f"out_{d}"], name=f"base_dsn out_{d}")
        so = torch.sigmoid(o)
        loss = loss + R.bce_loss(so, lab) + R.dice_loss(so, lab)
    close(float(loss), g["loss"], name="base_dsn loss")
    loss.backward()
    names, norms, _ = grad_digest({k: v.grad for k, v in leaves.items()})
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 2e-3, f"{names[int(rel.argmax())]} {rel.max():.2e}"
    close([float(P[str(k)].double().sum()) for k in g["bn_names"]], g["bn_sums"], name="bn running sums per domain")


def _factoratt_params(tag_shape, seed=4):
    """Rebuild the module-local weights gen_golden.gen_factoratt used (sorted named_parameters order)."""
    B, H, W, C = tag_shape
    Ch, hid = C // 8, max(C // 2, 4)
    shapes = {
        "crpe.conv_list.0.bias": (2 * Ch,), "crpe.conv_list.0.weight": (2 * Ch, 1, 3, 3),
        "crpe.conv_list.1.bias": (3 * Ch,), "crpe.conv_list.1.weight": (3 * Ch, 1, 5, 5),
        "crpe.conv_list.2.bias": (3 * Ch,), "crpe.conv_list.2.weight": (3 * Ch, 1, 7, 7),
        "domain_layer.0.bias": (hid,), "domain_layer.0.weight": (hid, 4),
        "domain_layer.2.bias": (C,), "domain_layer.2.weight": (C, hid),
        "proj.bias": (C,), "proj.weight": (C, C), "qkv.bias": (3 * C,), "qkv.weight": (3 * C, C),
    }
    P = {}
    for k, n in enumerate(sorted(shapes), start=1):
        shp = shapes[n]
        scale = 1.5 if "domain_layer" in n else (0.1 if n.endswith("bias") else (3.0 / shp[1]) ** 0.5 if len(shp) == 2 else 0.3)
        P[n] = synth_tokens(seed, 1000 + k, shp) * scale
    return P


@pytest.mark.parametrize("tag", ["c64", "c128", "c320"])
def test_factoratt_sup(golden, tag):
    g = golden("factoratt_small")
    B, H, W, C = [int(v) for v in g[f"{tag}_shape"]]
    raw = _factoratt_params((B, H, W, C))
    P = {("att." + k if not k.startswith("crpe") else k): v.clone().requires_grad_(True) for k, v in raw.items()}
    x = synth_tokens(4, 1, (B, H * W, C)).requires_grad_(True)
    dl = F.one_hot(torch.tensor([1, 3][:B]), 4).float()
    y = R.factor_att(P, "att", "crpe", x, H, W, 8, dl, R.RefState(training=True))
    close(y, g[f"{tag}_y"], name="y")
    gy = synth_tokens(4, 2, tuple(y.shape))
    (y * gy).sum().backward()
    close(x.grad, g[f"{tag}_dx"], rtol=1e-3, name="dx")
    for key in g.files:
        if key.startswith(f"{tag}_grad::"):
            n = key.split("::")[1]
            close(P[n if n.startswith("crpe") else "att." + n].grad, g[key], rtol=1e-3, name=key)
    assert bool(g[f"{tag}_da_gather_bitexact"])


def test_losses(golden):
    g = golden("losses_small")
    o = synth_tokens(5, 1, (2, 1, 32, 32)) * 6.0
    a = synth_tokens(5, 2, (2, 1, 32, 32)) * 6.0
    o.view(-1)[:8] = torch.tensor([200.0, -200.0, 120.0, -120.0, 90.0, -90.0, 40.0, -40.0])
    lab = synth_label(5, 2, 32, 32)
    o.requires_grad_(True); a.requires_grad_(True)
    l, la, lk = R.domain_losses(o, a, lab)
    close([float(l), float(la), float(lk)], g["losses"], name="losses")
    la.backward(retain_graph=True)
    close(a.grad, g["d_aux_from_auxloss"], rtol=1e-4, name="d aux / aux loss")
    a.grad = None
    (0.5 * lk + 0.5 * l).backward()
    close(o.grad, g["d_out_from_uni"], rtol=1e-4, name="d out / uni")
    close(a.grad, g["d_aux_from_uni"], rtol=1e-4, name="d aux / uni")


def test_metric_and_loader_restatements_known_answers():
    """oracle/pipeline.py (medpy dc / jc, the loader's normalisation) on hand-computed cases"""
    import numpy as np
    from oracle import pipeline as P
    a = np.array([[1, 1, 0, 0], [1, 0, 0, 0]]); b = np.array([[1, 0, 0, 0], [1, 1, 0, 1]])
    assert P.dc(a, b) == 2 * 2 / (3 + 4) and P.jc(a, b) == 2 / 5
    z = np.zeros((2, 2))
    assert P.dc(z, z) == 0.0 and P.jc(z, z) == 0.0
    img = np.array([[[0, 128, 255]]], dtype=np.uint8)
    t = P.load_image(img)
    want = [(np.float32(0 / 255) - np.float32(0.485)) / np.float32(0.229), (np.float32(128 / 255) - np.float32(0.456)) / np.float32(0.224),
            (np.float32(255 / 255) - np.float32(0.406)) / np.float32(0.225)]
    assert t.shape == (3, 1, 1) and [float(v) for v in t.view(-1)] == [float(w) for w in want]


def test_mdvit_dsn_two_sweep_step(golden):
    """MDViT_DSN (per-domain norms selected by int(d)): the oracle's renamed-view restatement vs the real reference"""
    from oracle.params import alias_map, param_spec
    g = golden("mdvit_dsn_step_64")
    S, B, seed = [int(v) for v in g["meta"]]
    pn = make_params(seed, model="MDViT_DSN", adapt_method="Sup")
    assert len(pn) + len(alias_map("MDViT")) == int(g["n_state_dict_keys"])       # unique + shared cpe/crpe aliases == reference keys
    P = R.to_torch(pn)
    batches = [(synth_image(500 + d, B, S, S), synth_label(600 + d, B, S, S), d) for d in range(4)]
    P2 = {k: v.clone() for k, v in P.items()}
    st = R.RefState(training=True)
    for d, (img, lab, sid) in enumerate(batches):
        dl = F.one_hot(torch.full((B,), sid, dtype=torch.long), 4).float()
        with torch.no_grad():
            o, a = R.mdvit_dsn_forward(P2, img, dl, str(sid), st)
            l = R.domain_losses(o, a, lab)
        close(o.reshape(-1)[::7], g[f"out_{d}"], name=f"out_{d}")
        close(a.reshape(-1)[::7], g[f"aux_{d}"], name=f"aux_{d}")
        close([float(v) for v in l], g[f"losses_{d}"], name=f"losses_{d}")
    bn_names = [str(n) for n in g["bn_names"]]
    close([float(P2[k].double().sum()) for k in bn_names], g["bn_sums"], name="bn running sums (only the domain's own norms move)")
    losses, grads = R.mdvit_train_step(P, batches, R.RefState(training=True), forward=R.mdvit_dsn_forward)
    close([losses["loss"], losses["aux_loss"], losses["kt_loss"]], g["total_losses"], name="total losses")
    names, norms, heads = grad_digest(grads)
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 2e-3, f"grad norm mismatch {names[int(rel.argmax())]} {rel.max():.2e}"
    for key in g.files:
        if key.startswith("grad::"):
            close(grads[key[6:]], g[key], rtol=1e-3, name=key)


def test_transfuse_oracle_vs_golden(golden):
    """oracle/transfuse_ref.py (TransFuse_S_adapt + structure_loss restated, ResNet-34 restated) against the fixture the reference's own
    TransFuse_S_adapt produced: 630 state_dict names, the three logit maps of two domains, the step losses, the norm and head of
    every one of the gradient tensors, BatchNorm running statistics."""
    from oracle import transfuse_ref as T
    from oracle.gen_golden import synth_image, synth_label, grad_digest
    g = golden("transfuse_step_256")
    S, B, seed = [int(v) for v in g["meta"]]
    pn = T.make_params(seed)
    assert len(pn) == int(g["n_state_dict_keys"]) == 630
    assert abs(sum(v.size for k, v in pn.items() if v.dtype.kind == "f" and "running_" not in k) - 26.873877e6) < 1
    P = T.to_torch(pn)
    batches = [(synth_image(1200 + d, B, S, S), synth_label(1300 + d, B, S, S), d) for d in (1, 3)]
    with torch.no_grad():
        for img, lab, d in batches:
            dl = torch.nn.functional.one_hot(torch.full((B,), d, dtype=torch.long), 4).float()
            m4, m3, m2 = T.transfuse_forward({k: v.clone() for k, v in P.items()}, img, dl, T.TFState(training=True))
            for nm, t in (("map_x", m4), ("map_1", m3), ("map_2", m2)):
                ref = g[f"{nm}_{d}"]
                got = t.reshape(-1)[::61].numpy()
                assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6), (nm, d)
    losses, grads = T.transfuse_train_step(P, batches, T.TFState(training=True))
    for l, d in zip(losses, (1, 3)):
        assert abs(l - float(g[f"loss_{d}"])) <= 2e-4 * abs(float(g[f"loss_{d}"]))
    names, norms, heads = grad_digest(grads)
    assert list(names) == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 5e-3, (names[int(rel.argmax())], float(rel.max()))
    for k in ("resnet.bn1.running_mean", "up_c.residual.bn1.running_var", "up_c_2_2.attn_block.psi.1.running_mean"):
        assert np.abs(P[k].numpy() - g["buf__" + k]).max() <= 2e-4 * max(np.abs(g["buf__" + k]).max(), 1e-6), k
