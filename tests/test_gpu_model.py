"""Module- and model-level parity of the HIP path (through the C ABI) against
  (1) the committed golden vectors the real reference produced (tests/golden/*.npz), and
  (2) the CPU oracle on the same seeded inputs,
plus the drop-in surface (state_dict names, two backward() calls on one graph, requires_grad flips)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3          # north_star: logits and losses within 1e-3 relative of the fp32 reference


def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def check(a, b, tol=TOL, name="", floor=0.0):
    e = relerr(a, b)
    if floor > 0.0:      # tensors that are analytically zero (e.g. a conv bias in front of train-mode BN) hold only round-off
        a_ = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
        b_ = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
        if float(a_.abs().max()) <= floor and float(b_.abs().max()) <= floor:
            return
    assert np.isfinite(e) and e <= tol, f"{name}: rel-to-max error {e:.3e} > {tol}"


GRAD_FLOOR = 1e-7       # |grad| below this in BOTH implementations == "zero up to round-off"


def check_grad(a, b, name="", l2_tol=3e-3, max_tol=6e-2):
    """Gradient parity that survives an isolated ReLU/Hardswish mask flip.

    A step pushes ~1e6 BatchNorm outputs through ReLU; the closest one to the kink is typically 1e-7..1e-6
    away (oracle.mdvit_ref.kink_margin measures it: 7.5e-9 for the harness data below), i.e. inside fp32
    round-off, so two correct fp32 implementations can disagree on THAT element's mask.  One flipped element
    moves one channel of the following BatchNorm/conv gradients by ~1/(tokens) -- up to a few % of the
    tensor's max for that channel -- while a genuine bug moves the whole tensor.  So: relative L2 error
    <= 3e-3 over the tensor AND no element off by more than 6 % of the tensor's max."""
    a_ = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b_ = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    if float(a_.abs().max()) <= GRAD_FLOOR and float(b_.abs().max()) <= GRAD_FLOOR:
        return
    l2 = float((a_ - b_).norm() / max(float(b_.norm()), 1e-30))
    mx = float((a_ - b_).abs().max() / max(float(b_.abs().max()), 1e-30))
    assert np.isfinite(l2) and l2 <= l2_tol and mx <= max_tol, f"{name}: rel L2 error {l2:.3e} (tol {l2_tol}), max-abs/max {mx:.3e} (tol {max_tol})"


def load_params(model, params_np):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in params_np.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    from oracle.params import alias_map
    am = alias_map(decoder_name="Transformer")          # (a superset of the other variants' aliases)
    assert not unexpected, unexpected[:5]
    assert all(k in am for k in missing), [k for k in missing if k not in am][:5]
    return model


def build_mdvit(seed, img_size=64, drop=0.0, decoder_name="MLPFM"):
    import mdvit_amd
    from oracle.params import make_params
    m = mdvit_amd.MDViT(img_size=img_size, drop_rate=drop, drop_path_rate=drop, conv_norm=torch.nn.BatchNorm2d,
                        adapt_method="Sup", num_domains=4, decoder_name=decoder_name)
    load_params(m, make_params(seed, model="MDViT", adapt_method="Sup", decoder_name=decoder_name))
    if drop == 0.0 and decoder_name == "DeepLabV3":
        for d in range(1, 5):
            getattr(m, f"debranch{d}").classifier[0].project[3].p = 0.0
    elif drop == 0.0 and decoder_name != "Transformer":
        for d in range(1, 5):
            getattr(m, f"debranch{d}").dropout.p = 0.0
    return m.to(dev())


def test_native_library_is_loaded():
    """the GPU suite must run on libmdvit_hip.so, not on any fallback"""
    from mdvit_amd import _lib
    lib = _lib.load()
    assert lib.mdvit_version() == 1
    with open("/proc/self/maps") as f:
        assert "libmdvit_hip.so" in f.read()


def test_state_dict_surface():
    import mdvit_amd
    from oracle.params import param_spec, alias_map
    m = mdvit_amd.MDViT(img_size=64, adapt_method="Sup")
    sd = m.state_dict()
    spec, am = param_spec("MDViT", "Sup"), alias_map()
    assert len(sd) == 608 and set(sd) == set(spec) | set(am)
    assert all(tuple(sd[k].shape) == tuple(s) for k, (_, s) in spec.items())
    assert len(dict(m.named_parameters())) == 432
    assert sum(1 for n, _ in m.named_parameters() if "domain_layer" in n) == 64


@pytest.mark.parametrize("tag", ["c64", "c128", "c320"])
def test_factoratt_module_vs_golden(golden, tag):
    """FactorAtt_ConvRelPosEnc_Sup fwd + all grads vs what the reference module produced (mdvit.py:243-313)."""
    from mdvit_amd.blocks import ConvRelPosEnc, FactorAtt_ConvRelPosEnc_Sup
    from oracle.gen_golden import synth_tokens
    from test_oracle_golden import _factoratt_params
    g = golden("factoratt_small")
    B, H, W, C = [int(v) for v in g[f"{tag}_shape"]]
    crpe = ConvRelPosEnc(Ch=C // 8, h=8, window={3: 2, 5: 3, 7: 3})
    att = FactorAtt_ConvRelPosEnc_Sup(H * W, C, num_heads=8, qkv_bias=True, shared_crpe=crpe, num_domains=4)
    raw = _factoratt_params((B, H, W, C))
    att.load_state_dict({k: v for k, v in raw.items()}, strict=True)
    att = att.to(dev()).train()
    x = synth_tokens(4, 1, (B, H * W, C)).to(dev()).requires_grad_(True)
    dl = F.one_hot(torch.tensor([1, 3][:B]), 4).float().to(dev())
    y = att(x, (H, W), dl)
    check(y, g[f"{tag}_y"], name="y")
    (y * synth_tokens(4, 2, tuple(y.shape)).to(dev())).sum().backward()
    check(x.grad, g[f"{tag}_dx"], name="dx")
    named = dict(att.named_parameters())
    for key in g.files:
        if key.startswith(f"{tag}_grad::"):
            check(named[key.split("::")[1]].grad, g[key], name=key)
        elif key.startswith(f"{tag}_gradsample::"):
            check(named[key.split("::")[1]].grad.reshape(-1)[::29], g[key], name=key)


@pytest.mark.parametrize("fixture,decoder_name", [("mdvit_step_64", "MLPFM"), ("mdvit_mlp_step_64", "MLP"),
                                                  ("mdvit_transformer_step_64", "Transformer"), ("mdvit_deeplab_step_64", "DeepLabV3"),
                                                  ("mdvit_deeplab_step_64_b4", "DeepLabV3")])
def test_mdvit_two_sweep_step_vs_golden(golden, gemm_precision, fixture, decoder_name):
    """4-domain step, multi_train_MDViT.py:129-207: logits, the three losses, BN running stats and every
    parameter gradient after the aux sweep (domain_layer frozen) + uni sweep.  decoder_name='MLP': the peer heads
    without the main decoder's feature (MLPDecoder, Decoders.py:239-286; mdvit.py:601-606)."""
    from mdvit_amd.losses import domain_losses
    from oracle.gen_golden import synth_image, synth_label, grad_digest
    g = golden(fixture)
    S, B, seed = [int(v) for v in g["meta"]]
    m = build_mdvit(seed, S, decoder_name=decoder_name).train()
    if "n_state_dict_keys" in g.files:
        assert len(m.state_dict()) == int(g["n_state_dict_keys"])
    tot = tot_aux = tot_kt = 0.0
    for d in range(4):
        img, lab = synth_image(100 + d, B, S, S).to(dev()), synth_label(200 + d, B, S, S).to(dev())
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
        out, aux = m(img, dl, str(d))
        assert out.shape == (B, 1, S, S) and aux.shape == (B, 1, S, S)
        check(out, g[f"out_{d}"], name=f"out_{d}")
        check(aux, g[f"aux_{d}"], name=f"aux_{d}")
        l, la, lk = domain_losses(out, aux, lab)
        check(torch.stack([l, la, lk]), g[f"losses_{d}"], name=f"losses_{d}")
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    sd = m.state_dict()
    check(torch.tensor([float(sd[str(k)].double().sum()) for k in g["bn_names"]]), g["bn_sums"], name="BN running stats")
    assert int(sd["stem.0.bn.num_batches_tracked"]) == 4
    m.zero_grad()
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    assert all(p.grad is None for n, p in m.named_parameters() if "domain_layer" in n)
    assert m.finalconv[0].weight.grad is None
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = True
    (0.5 * tot_kt + 0.5 * tot).backward()
    grads = {n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in m.named_parameters()}
    names, norms, heads = grad_digest(grads)
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    # The ASPP pooling branch of the DeepLabV3 heads (Utils/_deeplab.py:124-135) normalises the B pooled vectors of a domain batch with a train-mode BatchNorm.
    # With B = 2 that backward divides by a two-sample variance and amplifies an fp32-ulp change of its input by ~1e5 (measured in round 5: re-ordered k slots and a
    # re-associated x * Phi(x), 1e-7 relative in the forward, moved that branch's parameter-gradient norms from 0.9 % to 3.5 % off the reference's own fp32 result):
    # the B = 2 fixture pins the forward, the losses and the running statistics, and says nothing about gradients in the bf16x3 mode.  The gradients are held
    # against the CONDITIONED fixture (`mdvit_deeplab_step_64_b4`: four samples per BatchNorm) at the bounds of the other head families.
    ill_conditioned = decoder_name == "DeepLabV3" and B < 4 and gemm_precision == "bf16x3"
    norm_tol = np.full(len(names), 5e-3)
    if decoder_name == "DeepLabV3" and gemm_precision == "bf16x3":
        norm_tol[:] = 1.5e-2
    if not ill_conditioned:
        worst = int((rel / norm_tol).argmax())
        assert (rel < norm_tol).all(), f"grad norm mismatch at {names[worst]}: {rel[worst]:.2e} (ours {norms[worst]:.4e} ref {ref[worst]:.4e}); next: " + \
            ", ".join(f"{names[i]} {rel[i]:.1e}" for i in np.argsort(-rel / norm_tol)[1:5])
        for key in g.files:
            if key.startswith("grad::"):      # 4-domain sums; this fixture's kink margin is 1.9e-6 (a flip is likely somewhere)
                check_grad(grads[key[6:]], g[key], name=key, l2_tol=1e-2)
    else:
        assert np.isfinite(norms).all() and (rel < 0.5).all()          # (same gradient up to the fixture's conditioning: finite and of the reference's size)


def test_mdvit_dsn_two_sweep_step_vs_golden(golden):
    """MDViT_DSN (domain-specific norms, mdvit.py:735-960): 4-domain two-sweep step against the real reference's
    fixture -- logits, losses, the per-domain running statistics (only the forward's own norms move), gradients of
    every norm bank, and the train step harness (per-domain forwards, and the domain-batched forward with the
    group-indexed norm kernels)"""
    import mdvit_amd
    from mdvit_amd.losses import domain_losses
    from mdvit_amd.train import mdvit_train_step
    from oracle.gen_golden import synth_image, synth_label, grad_digest
    from oracle.params import make_params
    g = golden("mdvit_dsn_step_64")
    S, B, seed = [int(v) for v in g["meta"]]

    def build():
        m = mdvit_amd.MDViT_DSN(img_size=S, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                                num_domains=4, decoder_name="MLPFM")
        load_params(m, make_params(seed, model="MDViT_DSN", adapt_method="Sup"))
        for d in range(1, 5):
            getattr(m, f"debranch{d}").dropout.p = 0.0
        return m.to(dev()).train()

    m = build()
    assert len(m.state_dict()) == int(g["n_state_dict_keys"])
    batches = [(synth_image(500 + d, B, S, S).to(dev()), synth_label(600 + d, B, S, S).to(dev()), torch.full((B,), d, dtype=torch.long))
               for d in range(4)]
    tot = tot_aux = tot_kt = 0.0
    for d, (img, lab, sid) in enumerate(batches):
        dl = F.one_hot(sid, 4).float().to(dev())
        out, aux = m(img, dl, str(d))
        check(out.reshape(-1)[::7], g[f"out_{d}"], name=f"out_{d}")
        check(aux.reshape(-1)[::7], g[f"aux_{d}"], name=f"aux_{d}")
        l, la, lk = domain_losses(out, aux, lab)
        check(torch.stack([l, la, lk]), g[f"losses_{d}"], name=f"losses_{d}")
        tot, tot_aux, tot_kt = tot + l, tot_aux + la, tot_kt + lk
    sd = m.state_dict()
    check(torch.tensor([float(sd[str(k)].double().sum()) for k in g["bn_names"]]), g["bn_sums"], name="BN running stats per domain")
    assert int(sd["stem_1.bns.2.num_batches_tracked"]) == 1
    m.zero_grad()
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = False
    tot_aux.backward(retain_graph=True)
    for n, p in m.named_parameters():
        if "domain_layer" in n:
            p.requires_grad = True
    (0.5 * tot_kt + 0.5 * tot).backward()
    grads = {n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in m.named_parameters()}
    names, norms, heads = grad_digest(grads)
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    worst = int(rel.argmax())
    assert rel.max() < 1e-2, f"grad norm mismatch at {names[worst]}: {rel.max():.2e} (ours {norms[worst]:.4e} ref {ref[worst]:.4e})"
    def bank_tol(name):
        # a per-domain norm bank row is trained by ONE domain's 2 images (a quarter of the step's samples): one flipped ReLU / Hardswish
        # derivative ("gradient metric") weighs four times as much in it as in a shared tensor
        return 2e-2 if any(t in name for t in ("norm1s.", "norm2s.", ".norms.", ".bns.", ".lns.")) else 1e-2
    for key in g.files:
        if key.startswith("grad::"):
            check_grad(grads[key[6:]], g[key], name=key, l2_tol=bank_tol(key))
    # the step harness: merged sweeps, per-domain forwards
    m2 = build()
    res = mdvit_train_step(m2, batches, optimizer=None, merged_sweeps=True)
    check(torch.stack([res["loss"], res["aux_loss"], res["kt_loss"]]), g["total_losses"], name="step losses")
    # the domain-batched step: ONE forward, norm bank row d on batch group d -- same losses, gradients, running statistics
    m3 = build()
    res3 = mdvit_train_step(m3, batches, optimizer=None, merged_sweeps=True, fuse_domains=4)
    check(torch.stack([res3["loss"], res3["aux_loss"], res3["kt_loss"]]), g["total_losses"], name="domain-batched step losses")
    sd3 = m3.state_dict()
    check(torch.tensor([float(sd3[str(k)].double().sum()) for k in g["bn_names"]]), g["bn_sums"], name="BN running stats per domain (batched)")
    assert int(sd3["stem_1.bns.2.num_batches_tracked"]) == 1
    g2 = dict(m2.named_parameters())
    for n, p in m3.named_parameters():
        assert (p.grad is None) == (g2[n].grad is None), n
        if p.grad is not None:
            check_grad(p.grad, g2[n].grad, name=f"batched vs per-domain {n}", l2_tol=3e-4, max_tol=3e-3)
    for key in g.files:
        if key.startswith("grad::"):
            check_grad(m3.get_parameter(key[6:]).grad.cpu(), g[key], name="batched " + key, l2_tol=bank_tol(key))
    # permuted / partial domain lists select the matching bank rows; repeated ids are refused
    img2 = torch.cat([batches[3][0], batches[1][0]]); lab2 = F.one_hot(torch.tensor([3] * B + [1] * B), 4).float().to(dev())
    m4 = build()
    out2, aux2 = m4(img2, lab2, ["3", "1"])
    check(out2[:B].reshape(-1)[::7], g["out_3"], name="batched ['3','1'] -> out_3")
    check(aux2[B:].reshape(-1)[::7], g["aux_1"], name="batched ['3','1'] -> aux_1")
    assert int(m4.state_dict()["stem_1.bns.3.num_batches_tracked"]) == 1 and int(m4.state_dict()["stem_1.bns.0.num_batches_tracked"]) == 0
    with pytest.raises(ValueError):
        m4(img2, lab2, ["1", "1"])
    # eval: the running statistics of each group's own bank
    m4.eval()
    with torch.no_grad():
        e_b = m4(img2, lab2, ["3", "1"])[0]
        e_s = torch.cat([m4(img2[:B], lab2[:B], "3")[0], m4(img2[B:], lab2[B:], "1")[0]])
    check(e_b, e_s, tol=1e-5, name="eval batched vs per-domain")


def test_mdvit_eval_vs_golden(golden):
    from oracle.gen_golden import synth_image
    g = golden("mdvit_eval_64")
    S, B, seed = [int(v) for v in g["meta"]]
    m = build_mdvit(seed, S).eval()
    with torch.no_grad():
        for d in (0, 3):
            dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
            out, aux = m(synth_image(300 + d, B, S, S).to(dev()), dl, str(d))
            check(out, g[f"out_{d}"], name="eval out")
            check(aux, g[f"aux_{d}"], name="eval aux")
        out, aux = m(synth_image(300, B, S, S).to(dev()), dl, "7")       # unknown d -> no aux head (mdvit.py:723-724)
        assert aux is None


def test_mdvit_rect_vs_golden(golden):
    from oracle.gen_golden import synth_image
    g = golden("mdvit_fwd_96x128")
    H, W, B, seed = [int(v) for v in g["meta"]]
    m = build_mdvit(seed, 128).train()
    with torch.no_grad():
        out, aux = m(synth_image(400, B, H, W).to(dev()), F.one_hot(torch.tensor([1]), 4).float().to(dev()), "1")
    check(out, g["out"], name="rect out")
    check(aux, g["aux"], name="rect aux")


def test_base_step_vs_golden(golden):
    import mdvit_amd
    from mdvit_amd.losses import seg_loss
    from oracle.gen_golden import synth_image, synth_label, grad_digest
    from oracle.params import make_params
    g = golden("base_step_64")
    S, B, seed = [int(v) for v in g["meta"]]
    m = mdvit_amd.BASE(drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method=False)
    load_params(m, make_params(seed, model="BASE", adapt_method=False))
    m = m.to(dev()).train()
    out = m(synth_image(500, B, S, S).to(dev()))
    check(out, g["out"], name="base out")
    loss = seg_loss(out, synth_label(600, B, S, S).to(dev()))
    check(loss, g["loss"], name="base loss")
    loss.backward()
    names, norms, _ = grad_digest({n: p.grad.detach().cpu() for n, p in m.named_parameters()})
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 5e-3, f"{names[int(rel.argmax())]} {rel.max():.2e}"


def test_base_dsn_step_vs_golden(golden):
    """BASE_DSN (base.py:515-700): per-domain norm banks under BASE's forward; also as ONE domain-batched forward"""
    import mdvit_amd
    from mdvit_amd.losses import seg_loss
    from oracle.gen_golden import synth_image, synth_label, grad_digest
    from oracle.params import make_params
    g = golden("base_dsn_step_64")
    S, B, seed = [int(v) for v in g["meta"]]

    def build():
        m = mdvit_amd.BASE_DSN(drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4)
        load_params(m, make_params(seed, model="BASE_DSN", adapt_method="Sup"))
        return m.to(dev()).train()

    m = build()
    assert len(m.state_dict()) == int(g["n_state_dict_keys"])
    data = {d: (synth_image(700 + d, B, S, S).to(dev()), synth_label(800 + d, B, S, S).to(dev()),
                F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())) for d in (2, 0)}
    loss = 0.0
    for d in (2, 0):
        out = m(data[d][0], data[d][2], str(d))
        check(out, g[f"out_{d}"], name=f"base_dsn out_{d}")
        loss = loss + seg_loss(out, data[d][1])
    check(loss, g["loss"], name="base_dsn loss")
    loss.backward()
    names, norms, _ = grad_digest({n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in m.named_parameters()})
    assert names == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    assert rel.max() < 5e-3, f"{names[int(rel.argmax())]} {rel.max():.2e}"
    sd = m.state_dict()
    check(torch.tensor([float(sd[str(k)].double().sum()) for k in g["bn_names"]]), g["bn_sums"], name="BN running stats per domain")
    # one domain-batched forward over both domains: bank row d on batch group d
    m2 = build()
    outb = m2(torch.cat([data[2][0], data[0][0]]), torch.cat([data[2][2], data[0][2]]), ["2", "0"])
    check(outb[:B], g["out_2"], name="batched out_2")
    check(outb[B:], g["out_0"], name="batched out_0")
    (seg_loss(outb[:B], data[2][1]) + seg_loss(outb[B:], data[0][1])).backward()
    for (n, p), (_, q) in zip(m2.named_parameters(), m.named_parameters()):
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            check_grad(p.grad, q.grad, name="batched " + n, l2_tol=1e-3, max_tol=1e-2)


@pytest.mark.parametrize("fuse,decoder_name", [(1, "MLPFM"), (4, "MLPFM"), (4, "DeepLabV3")])
def test_no_kernel_reads_an_unwritten_buffer(fuse, decoder_name):
    """MDVIT_POISON=1 NaN-fills every buffer an op allocates; a forward + backward must stay NaN-free.  (Caught a temporary W^T
    of a composed -- non-leaf -- weight being freed before the split-K data-gradient GEMM allocated its workspace over it:
    wrong gradients only when the allocator happened to reuse that block.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "debug_poison.py"), str(fuse), decoder_name],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MDVIT_POISON="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "forward NaN: False False" in r.stdout and "\n0 gradients with NaN" in r.stdout, r.stdout[-2000:]
    if fuse == 4 and decoder_name == "MLPFM":      # and the whole bench-style step: side-stream weight gradients into the buckets, fused AdamW
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "debug_poison_step.py")], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MDVIT_POISON="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "0 non-finite parameters" in r.stdout and "0 non-finite buffers" in r.stdout, r.stdout[-2000:]


def test_forty_steps_on_one_batch_drive_the_losses_down():
    """end to end: domain-batched forward, merged sweeps, side-stream weight gradients into the buckets, one-launch AdamW"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "overfit_check.py"), "64", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and "ok: loss" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_mdvit_vs_oracle_128(gemm_precision):
    """same seeded inputs, larger image (128x128), HIP path vs the CPU oracle incl. input-side gradients of
    every parameter (full tensors, not digests)."""
    from mdvit_amd.losses import domain_losses
    from oracle import mdvit_ref as R
    from oracle.gen_golden import synth_image, synth_label
    from oracle.params import make_params
    S, B, d = 128, 1, 2
    pn = make_params(11, model="MDViT", adapt_method="Sup")
    img, lab = synth_image(700, B, S, S), synth_label(701, B, S, S)
    losses, grads = R.mdvit_train_step(R.to_torch(pn), [(img, lab, d)], R.RefState(training=True))
    m = build_mdvit(11, S).train()
    dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
    out, aux = m(img.to(dev()), dl, str(d))
    l, la, lk = domain_losses(out, aux, lab.to(dev()))
    check(torch.stack([l, la, lk]), [losses["loss"], losses["aux_loss"], losses["kt_loss"]], name="losses")
    da = [p for n, p in m.named_parameters() if "domain_layer" in n]
    for p in da:
        p.requires_grad = False
    la.backward(retain_graph=True)
    for p in da:
        p.requires_grad = True
    (0.5 * lk + 0.5 * l).backward()
    bad = []
    for n, p in m.named_parameters():
        ref = grads[n]
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        # B = 1 at 128x128: the deepest BatchNorms normalise over 16 samples, the worst-conditioned case in the suite.
        # fp32 GEMMs: 3e-3 (mask flips only).  bf16x3 GEMMs perturb the forward by ~1e-5, i.e. a few more flips: 1e-2.
        # The bridge sits at 4x4 tokens x 1 image: its BatchNorms see 16 samples, so ONE flipped ReLU moves a channel's
        # gradient by ~1/16 -- under bf16x3 a couple of such flips happen; bound them separately (fp32: no exception).
        l2_tol, max_tol = (3e-3, 6e-2) if gemm_precision == "fp32" else ((5e-2, 0.5) if n.startswith("bridge.") else (1e-2, 6e-2))
        try:
            check_grad(p.grad, ref, name=n, l2_tol=l2_tol, max_tol=max_tol)
        except AssertionError as exc:
            bad.append(str(exc))
    assert not bad, f"{len(bad)} gradient tensors off: {bad[:6]}"


def test_train_mode_dropout_runs_and_varies():
    """drop_rate = drop_path_rate = 0.1 (the reference's setting): finite, stochastic, eval deterministic."""
    from oracle.gen_golden import synth_image
    torch.manual_seed(0)
    m = build_mdvit(0, 64, drop=0.1).train()
    img = synth_image(800, 4, 64, 64).to(dev())
    dl = F.one_hot(torch.zeros(4, dtype=torch.long), 4).float().to(dev())
    o1, a1 = m(img, dl, "0")
    o2, a2 = m(img, dl, "0")
    assert torch.isfinite(o1).all() and torch.isfinite(a1).all()
    assert (o1 != o2).any() and (a1 != a2).any()
    (o1.sum() + a1.sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(img, dl, "0")[0], m(img, dl, "0")[0]
    assert torch.equal(e1, e2)


def test_train_step_harness_matches_reference_order():
    """per-domain backward (memory-lean) == the reference's 4-forwards-then-2-sweeps order."""
    from mdvit_amd.train import mdvit_train_step
    from oracle.gen_golden import synth_image, synth_label
    batches = [(synth_image(900 + d, 2, 64, 64).to(dev()), synth_label(910 + d, 2, 64, 64).to(dev()),
                torch.full((2,), d, dtype=torch.long, device=dev())) for d in range(4)]
    res = []
    for per_domain in (True, False):
        m = build_mdvit(5, 64).train()
        out = mdvit_train_step(m, batches, optimizer=None, per_domain_backward=per_domain)
        res.append((out, {n: p.grad.clone() for n, p in m.named_parameters()}))
    for k in ("loss", "aux_loss", "kt_loss"):
        check(res[0][0][k], res[1][0][k], tol=1e-5, name=k)
    for n in res[0][1]:
        check_grad(res[0][1][n], res[1][1][n], name=n)


def test_forward_is_bitwise_reproducible():
    """channel statistics are reduced in a fixed order: two identical train-mode forwards agree bit for bit
    (needed because an activation that sits on the ReLU kink would otherwise flip its gradient mask run to run)"""
    from oracle.gen_golden import synth_image
    img = synth_image(950, 2, 64, 64).to(dev())
    dl = F.one_hot(torch.tensor([2, 2]), 4).float().to(dev())
    outs = []
    for _ in range(3):
        m = build_mdvit(7, 64).train()
        with torch.no_grad():
            o, a = m(img, dl, "2")
        outs.append((o.clone(), a.clone(), m.stem[0].bn.running_var.clone()))
    for o, a, rv in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(a, outs[0][1]) and torch.equal(rv, outs[0][2])


def test_merged_sweeps_equal_reference_two_sweeps():
    """dgrad-only aux sweep + one merged sweep == the reference's two full sweeps (every parameter, incl. the
    domain adapters that the reference freezes during the aux sweep)"""
    from mdvit_amd.train import mdvit_train_step
    from oracle.gen_golden import synth_image, synth_label
    batches = [(synth_image(960 + d, 2, 64, 64).to(dev()), synth_label(970 + d, 2, 64, 64).to(dev()),
                torch.full((2,), d, dtype=torch.long, device=dev())) for d in range(4)]
    res = []
    for merged in (False, True):
        m = build_mdvit(9, 64).train()
        out = mdvit_train_step(m, batches, optimizer=None, merged_sweeps=merged)
        res.append((out, {n: p.grad.clone() for n, p in m.named_parameters()}))
        assert all(p.requires_grad for p in m.parameters())
    for k in ("loss", "aux_loss", "kt_loss"):
        check(res[0][0][k], res[1][0][k], tol=1e-6, name=k)
    for n in res[0][1]:
        check_grad(res[1][1][n], res[0][1][n], name=n, l2_tol=2e-4, max_tol=2e-3)


def test_side_stream_wgrad_matches_single_stream():
    """weight gradients issued on the side HIP stream (joined before use) == everything on one stream"""
    from mdvit_amd import ops
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.train import mdvit_train_step
    from oracle.gen_golden import synth_image, synth_label
    batches = [(synth_image(980 + d, 2, 64, 64).to(dev()), synth_label(990 + d, 2, 64, 64).to(dev()),
                torch.full((2,), d, dtype=torch.long, device=dev())) for d in range(4)]
    res = []
    try:
        for side in (False, True):
            ops.enable_side_stream(side)
            m = build_mdvit(13, 64).train()
            acc = GradAccumulator(m.parameters())
            acc.attach_sinks(side)      # side run: wgrad GEMMs accumulate straight into the buckets on the side stream
            for _ in range(2):          # twice: exercises buffer reuse across steps
                mdvit_train_step(m, batches, optimizer=None, accumulator=acc, merged_sweeps=True)
            torch.cuda.synchronize()
            res.append({n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        ops.enable_side_stream(False)
        ops.set_grad_sinks(None)
    for n in res[0]:
        check_grad(res[1][n], res[0][n], name=n, l2_tol=2e-4, max_tol=2e-3)


def _four_domain_batches(seed, B=2, size=64):
    from oracle.gen_golden import synth_image, synth_label
    return [(synth_image(seed + d, B, size, size).to(dev()), synth_label(seed + 10 + d, B, size, size).to(dev()),
             torch.full((B,), d, dtype=torch.long)) for d in range(4)]


def test_train_step_metrics_on_device_match_host_restatement():
    """with_metrics=True: per-domain Dice / IoU of the thresholded logits, computed on the device inside the step, equal
    medpy's dc / jc (oracle/pipeline.py) of the same logits taken to the host -- for fused and per-domain forwards"""
    from mdvit_amd.train import mdvit_train_step
    from oracle import pipeline as P
    batches = _four_domain_batches(1400)
    m = build_mdvit(29, 64).eval()                 # eval: the two calls below see identical logits
    with torch.no_grad():
        logits = [m(b[0], F.one_hot(b[2], 4).float().to(dev()), str(d))[0] for d, b in enumerate(batches)]
    want = [P.train_metrics(lg.cpu(), b[1].cpu()) for lg, b in zip(logits, batches)]
    for fuse in (1, 4):
        out = mdvit_train_step(m, batches, optimizer=None, fuse_domains=fuse, with_metrics=True)
        got = out["metrics"].cpu()
        assert got.shape == (4, 4)
        for d in range(4):
            assert abs(float(got[d, 0]) - want[d][0]) < 1e-6 and abs(float(got[d, 1]) - want[d][1]) < 1e-6, (fuse, d, got[d], want[d])
        m.zero_grad(set_to_none=True)


@pytest.mark.parametrize("fuse,decoder_name", [(2, "MLPFM"), (4, "MLPFM"), (2, "MLP"), (4, "Transformer"), (4, "DeepLabV3")])
def test_domain_batched_step_equals_per_domain_forwards(fuse, decoder_name):
    """ONE forward over the concatenated domain batches (per-domain BatchNorm statistics, per-domain peer heads and
    losses) == the reference's one forward per domain: logits, the three losses, every gradient, BN running stats"""
    from mdvit_amd.train import mdvit_train_step
    batches = _four_domain_batches(1100)
    res = []
    for f in (1, fuse):
        m = build_mdvit(17, 64, decoder_name=decoder_name).train()
        out = mdvit_train_step(m, batches, optimizer=None, merged_sweeps=True, fuse_domains=f)
        res.append((out, {n: p.grad.clone() for n, p in m.named_parameters()}, {n: b.clone() for n, b in m.named_buffers()}))
    for k in ("loss", "aux_loss", "kt_loss"):
        check(res[0][0][k], res[1][0][k], tol=1e-5, name=k)
    # (the batched GEMMs take other tile / split-K plans than the per-domain ones: last-bit differences that can flip one
    #  ReLU / Hardswish mask of these 2-image BatchNorms -- see check_grad; a wrong group mapping would be off by O(1))
    for n in res[0][1]:
        check_grad(res[1][1][n], res[0][1][n], name=n, l2_tol=1e-3, max_tol=1e-2)
    for n in res[0][2]:
        check(res[1][2][n].double(), res[0][2][n].double(), tol=1e-5, name=n)


def test_domain_batched_forward_matches_oracle_per_domain():
    """model(x, label, ['0','1','2','3']) vs the oracle run once per domain"""
    from oracle import mdvit_ref as R
    from oracle.params import make_params
    P = make_params(21, model="MDViT", adapt_method="Sup")
    m = build_mdvit(21, 64).train()
    batches = _four_domain_batches(1200)
    img = torch.cat([b[0] for b in batches]); sid = torch.cat([b[2] for b in batches])
    lab = F.one_hot(sid, 4).float().to(dev())
    out, aux = m(img, lab, ["0", "1", "2", "3"])
    Pt = R.to_torch(P)
    for d, b in enumerate(batches):
        st = R.RefState(training=True)          # running statistics advance domain by domain, as in the fused forward
        with torch.no_grad():
            o_ref, a_ref = R.mdvit_forward(Pt, b[0].cpu(), F.one_hot(b[2], 4).float(), str(d), st)
        check(out[2 * d:2 * d + 2].cpu(), o_ref, tol=1e-3, name=f"logits d{d}")
        check(aux[2 * d:2 * d + 2].cpu(), a_ref, tol=1e-3, name=f"aux d{d}")
    sd = m.state_dict()
    for k in ("stem.0.bn.running_mean", "bridge.1.running_var", "decoder4.conv1.1.running_mean"):
        if k in sd and k in Pt:
            check(sd[k].cpu(), Pt[k], tol=1e-4, name=k)


def test_device_seed_redraws_dropout_masks(monkeypatch):
    """with the device-side seed enabled the SAME host keys (as baked into a captured graph) give a new mask after
    bump_seed(), the same mask without it, and the backward re-derives the mask of its forward"""
    from mdvit_amd import ops
    monkeypatch.setattr(ops, "_next_key", lambda: (123, 456))
    ops.enable_device_seed(True)
    try:
        x = torch.ones(512, 256, device=dev())
        w = torch.eye(256, device=dev()).requires_grad_(True)
        y0 = ops.linear(x, w, None, drop_p=0.5)
        y0b = ops.linear(x, w, None, drop_p=0.5)
        assert torch.equal(y0, y0b)
        ops.bump_seed()
        y1 = ops.linear(x, w, None, drop_p=0.5)
        for y in (y0, y1):
            assert abs((y != 0).float().mean().item() - 0.5) < 0.02
        assert (y0.detach() != y1.detach()).float().mean().item() > 0.3
        # backward of y1 uses y1's mask: d(sum y1)/dx is 2 where kept, 0 where dropped (w = I)
        xg = x.clone().requires_grad_(True)
        y2 = ops.linear(xg, w, None, drop_p=0.5)
        y2.sum().backward()
        assert torch.equal(xg.grad, y2.detach())
    finally:
        ops.enable_device_seed(False)


def test_graphed_step_matches_eager_step():
    """HIP-graph replay of the whole optimisation step (dropout off) == the eager step: same losses and weights after
    three steps"""
    from mdvit_amd import ops
    from mdvit_amd.graph import GraphedStep
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.train import mdvit_train_step
    batches = _four_domain_batches(1300)
    res = []
    try:
        for graphed in (False, True):
            m = build_mdvit(23, 64).train()
            # SGD: Adam's first steps move every weight by ~lr whatever its gradient, so run-to-run float-atomic noise in
            # near-zero gradients would turn into +-lr weight differences and hide what is being tested
            opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, foreach=True)
            acc = GradAccumulator(m.parameters())
            acc.attach_sinks()
            fn = lambda b: mdvit_train_step(m, b, optimizer=opt, accumulator=acc, merged_sweeps=True, fuse_domains=4)
            losses = []
            if graphed:
                g = GraphedStep(fn, batches, warmup=1, fuse_domains=4)     # the eager warm-up step is step 1
                losses.append(None)
                for _ in range(2):
                    losses.append(float(g(batches)["loss"]))
            else:
                for _ in range(3):
                    losses.append(float(fn(batches)["loss"]))
            torch.cuda.synchronize()
            res.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
    finally:
        ops.set_grad_sinks(None)
        ops.enable_device_seed(False)
    for a, b in zip(res[0][0][1:], res[1][0][1:]):
        assert abs(a - b) <= 2e-4 * abs(a), (res[0][0], res[1][0])
    for n in res[0][1]:
        check(res[1][1][n], res[0][1][n], tol=2e-3, name=n)


def test_graphed_base_step_matches_eager_step():
    """the BASE step (BASELINE configs[0]: what bench.py's base_bs4_gpu leg replays as a HIP graph) captured == eager"""
    import mdvit_amd
    from mdvit_amd import ops
    from mdvit_amd.graph import GraphedStep
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.train import base_train_step
    from oracle.gen_golden import synth_image, synth_label
    from oracle.params import make_params
    S, B = 64, 2
    batch = [(synth_image(1700, B, S, S).to(dev()), synth_label(1701, B, S, S).to(dev()), torch.zeros(B, dtype=torch.long))]
    res = []
    try:
        for graphed in (False, True):
            m = mdvit_amd.BASE(drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method=False)
            load_params(m, make_params(5, model="BASE", adapt_method=False))
            m = m.to(dev()).train()
            opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, foreach=True)
            acc = GradAccumulator(m.parameters())
            acc.attach_sinks()
            fn = lambda b: base_train_step(m, b, optimizer=opt, accumulator=acc)
            losses = []
            if graphed:
                g = GraphedStep(fn, batch, warmup=1, fuse_domains=1)
                losses.append(None)
                for _ in range(2):
                    losses.append(float(g(batch)["loss"]))
            else:
                for _ in range(3):
                    losses.append(float(fn(batch)["loss"]))
            torch.cuda.synchronize()
            res.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
    finally:
        ops.set_grad_sinks(None)
        ops.enable_device_seed(False)
    for a, b in zip(res[0][0][1:], res[1][0][1:]):
        assert abs(a - b) <= 2e-4 * abs(a), (res[0][0], res[1][0])
    for n in res[0][1]:
        check(res[1][1][n], res[0][1][n], tol=2e-3, name=n)


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29581"), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.parallel import GradAccumulator, broadcast_parameters
from mdvit_amd.synthetic import make_step_batches
from mdvit_amd.train import mdvit_train_step
res = []
# (collectives forced?, adapters in late buckets?, repeats).  The last configuration is bench.py's: every bucket but the adapters' goes on
# the wire (RCCL, issued from the side stream's context) when the FULL sweep ends and is reduced underneath the aux sweep, which runs on a
# stream of its own; repeated, because a missing cross-stream dependency shows as run-to-run drift.
for force, late, reps in ((False, False, 1), (True, False, 1), (True, True, 3)):
    ops._force_collectives = force            # RCCL all-reduce of the loss sums and of the gradient buckets, world = 1
    for rep in range(reps):
        torch.manual_seed(0)
        m = mdvit_amd.MDViT(img_size=64, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup",
                            num_domains=4, decoder_name="MLPFM").cuda().train()
        for d in range(1, 5):
            getattr(m, f"debranch{d}").dropout.p = 0.0
        broadcast_parameters(m)
        ops.enable_side_stream(True)
        da = [p for n, p in m.named_parameters() if "domain_layer" in n]
        acc = GradAccumulator(m.parameters(), bucket_bytes=4 << 20, late=da if late else None); acc.attach_sinks()
        b = make_step_batches(2, 64, rank=0, step=0, device=torch.device("cuda", 0))
        out = mdvit_train_step(m, b, optimizer=None, accumulator=acc, merged_sweeps=True, fuse_domains=4)
        torch.cuda.synchronize()
        if late:
            n_early = acc._n_early
            assert n_early > 1 and len(acc.reducer.buckets) > n_early, (n_early, len(acc.reducer.buckets))
            assert acc.overlapped_buckets == n_early, (acc.overlapped_buckets, n_early)        # all of them went on the wire under the aux sweep
        res.append(([float(out[k]) for k in ("loss", "aux_loss", "kt_loss")], [p.grad.clone() for p in m.parameters()]))
        ops.set_grad_sinks(None); ops.enable_side_stream(False)
for r in res[1:]:
    for a, b in zip(res[0][0], r[0]):
        assert abs(a - b) <= 1e-5 * abs(a), (res[0][0], r[0])
    worst = max(float((x - y).norm() / (y.norm() + 1e-20)) for x, y in zip(r[1], res[0][1]))
    assert worst < 2e-3, worst
# the three overlapped runs against each other: bitwise but for the float atomics of the window / depthwise weight gradients and the adapter's e
drift = max(float((x - y).abs().max() / (y.abs().max() + 1e-20)) for r in res[3:] for x, y in zip(r[1], res[2][1]))
assert drift < 1e-5, drift
dist.barrier(); dist.destroy_process_group()
print("dp-path ok", worst, drift)
"""


def test_data_parallel_code_path_on_one_gpu(tmp_path):
    """the collective code paths of the DP step (RCCL all-reduce of the 16 loss sums per domain and of the gradient buckets,
    side-stream weight gradients into the buckets) run in a 1-rank NCCL group and reproduce the plain step -- including the OVERLAPPED
    form bench.py runs: the adapters in `late=` buckets, every other bucket launched from the side stream's context when the full sweep
    ends (multi_train_MDViT.py:72-74's DataParallel replaced), three times over to catch a missing cross-stream dependency"""
    import subprocess, sys
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "dp-path ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


_DP2_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", device_id=dev)
import mdvit_amd
from mdvit_amd import ops
from mdvit_amd.parallel import GradAccumulator, broadcast_parameters
from mdvit_amd.synthetic import make_step_batches
from mdvit_amd.train import mdvit_train_step

def model():
    torch.manual_seed(0)
    m = mdvit_amd.MDViT(img_size=64, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                        decoder_name="MLPFM").to(dev).train()
    for d in range(1, 5):
        getattr(m, f"debranch{d}").dropout.p = 0.0
    return m

def dp_step(global_losses):
    ops.set_global_batch_losses(global_losses)
    ops._force_collectives = world == 1       # (a one-rank dry run of this script still takes the collective code paths)
    m = model()
    broadcast_parameters(m)
    ops.enable_side_stream(True)
    da = [p for n, p in m.named_parameters() if "domain_layer" in n]
    acc = GradAccumulator(m.parameters(), bucket_bytes=4 << 20, late=da); acc.attach_sinks()
    out = mdvit_train_step(m, make_step_batches(2, 64, rank=rank, step=0, device=dev), optimizer=None, accumulator=acc, merged_sweeps=True, fuse_domains=4)
    torch.cuda.synchronize()
    assert acc.overlapped_buckets == acc._n_early > 0, (acc.overlapped_buckets, acc._n_early)       # on the wire underneath the aux sweep
    g = [p.grad.clone() for p in m.parameters()]
    ops.set_grad_sinks(None); ops.enable_side_stream(False); ops._force_collectives = False
    return [float(out[k]) for k in ("loss", "aux_loss", "kt_loss")], g

# (1) per-rank losses: the DP gradient is the mean of the ranks' own gradients -- the reference is two plain single-process steps, no collective in them
l_dp, g_dp = dp_step(False)
ops.set_global_batch_losses(False)
ref = None
for r in range(world):
    m = model()
    mdvit_train_step(m, make_step_batches(2, 64, rank=r, step=0, device=dev), optimizer=None, merged_sweeps=True, fuse_domains=4)
    torch.cuda.synchronize()
    g = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in m.parameters()]
    ref = g if ref is None else [a + b for a, b in zip(ref, g)]
ref = [a / world for a in ref]
worst = max(float((x - y).norm() / (y.norm() + 1e-20)) for x, y in zip(g_dp, ref) if float(y.norm()) > 0)
assert worst < 2e-3, worst
# (2) global-batch losses (the 16 loss sums per domain all-reduced): every rank ends with the SAME averaged gradients and the SAME losses
l_g, g_g = dp_step(True)
flat = torch.cat([t.reshape(-1) for t in g_g]); other = flat.clone()
dist.broadcast(other, src=0)
assert torch.equal(flat, other) and bool(torch.isfinite(flat).all())
lt = torch.tensor(l_g, device=dev); l0 = lt.clone(); dist.broadcast(l0, src=0)
assert torch.equal(lt, l0), (lt, l0)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "dp2 ok", worst)
"""


def test_data_parallel_two_ranks_over_rccl(tmp_path):
    """World size 2 over RCCL (skipped below two devices -- the builder's pool has one; the first multi-GPU box runs it): the product DP step (gradient
    buckets with the adapters late, every other bucket all-reduced from the side stream's context underneath the aux sweep, loss sums all-reduced) against
    single-process references -- the DataParallel replacement of multi_train_MDViT.py:72-74."""
    import subprocess, sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "dp2_worker.py"
    script.write_text(_DP2_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} dp2 ok" in o, f"rank {r}:\n{o[-3000:]}"


# ------------------------------------------------------------------------------------------------------------------------
# round 2: the optimizer owns the invalidation of the cached W^T / weight planes; parity at the benchmark's own size
# ------------------------------------------------------------------------------------------------------------------------
def test_base_fused_adamw_second_step_gradients_follow_the_updated_weights():
    """FusedAdamW writes the parameters through raw pointers (Tensor._version does not move): the GEMMs' cached transposes and
    weight planes must still follow.  Two BASE steps through train.base_train_step + FusedAdamW; the gradients of step 2 are
    compared with the oracle evaluated at the weights step 1 produced (a stale cache shows up as a wrong dx everywhere)."""
    import mdvit_amd
    from mdvit_amd import ops
    from mdvit_amd.optim import FusedAdamW
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.train import base_train_step
    from oracle import mdvit_ref as R
    from oracle.gen_golden import synth_image, synth_label
    from oracle.params import make_params
    S, B = 64, 2
    m = mdvit_amd.BASE(drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method=False)
    load_params(m, make_params(3, model="BASE", adapt_method=False))
    m = m.to(dev()).train()
    acc = GradAccumulator(m.parameters()); acc.attach_sinks()
    try:
        opt = FusedAdamW(acc, lr=2e-2, weight_decay=0.0)           # a large step: stale weights would be far off
        img, lab = synth_image(1500, B, S, S), synth_label(1501, B, S, S)
        batch = [(img.to(dev()), lab.to(dev()), torch.zeros(B, dtype=torch.long))]
        base_train_step(m, batch, optimizer=opt, accumulator=acc)                       # step 1 (updates the weights)
        P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}            # the weights step 2 sees
        base_train_step(m, batch, optimizer=None, accumulator=acc)                      # step 2: gradients only
        torch.cuda.synchronize()
        got = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    finally:
        ops.set_grad_sinks(None)
    # the BN running statistics of P are those AFTER step 1's forward; the train-mode forward does not read them
    _, grads = R.base_train_step(P, img, lab, None, R.RefState(training=True))
    bad = []
    for n, g_ in got.items():
        ref = grads.get(n)
        if ref is None:
            continue
        try:
            check_grad(g_, ref, name=n, l2_tol=1e-2, max_tol=6e-2)
        except AssertionError as exc:
            bad.append(str(exc))
    assert not bad, f"{len(bad)} gradient tensors off after an optimizer step: {bad[:5]}"


def test_mdvit_512_one_image_vs_oracle(gemm_precision):
    """The benchmark's image size: MDViT Sup, one domain, one 512x512 image, HIP vs the CPU oracle -- logits and the three losses
    at 1e-3, every gradient tensor by relative L2 (the tile-count-, split-K- and planner-dependent kernel paths that 64x64 / 128x128
    inputs never reach: 128 attention tiles per image, the 128x128 / 256x64 GEMM plans, split-K weight gradients over 16384 tokens)."""
    from mdvit_amd.losses import domain_losses
    from oracle import mdvit_ref as R
    from oracle.gen_golden import synth_image, synth_label
    from oracle.params import make_params
    S, B, d = 512, 1, 3
    pn = make_params(17, model="MDViT", adapt_method="Sup")
    img, lab = synth_image(1700, B, S, S), synth_label(1701, B, S, S)
    m = build_mdvit(17, S).train()
    dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
    out, aux = m(img.to(dev()), dl, str(d))
    l, la, lk = domain_losses(out, aux, lab.to(dev()))
    da = [p for n, p in m.named_parameters() if "domain_layer" in n]
    for p in da:
        p.requires_grad = False
    la.backward(retain_graph=True)
    for p in da:
        p.requires_grad = True
    (0.5 * lk + 0.5 * l).backward()
    torch.cuda.synchronize()
    got_out, got_aux = out.detach().cpu(), aux.detach().cpu()
    got_grads = {n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in m.named_parameters()}
    got_losses = torch.stack([l, la, lk]).detach().cpu()
    del m, out, aux
    torch.cuda.empty_cache()
    P = R.to_torch(pn)
    st = R.RefState(training=True)
    with torch.no_grad():
        ro, ra = R.mdvit_forward({k: v.clone() for k, v in P.items()}, img, F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float(), str(d), st)
    check(got_out, ro, name="512 logits")
    check(got_aux, ra, name="512 aux logits")
    losses, grads = R.mdvit_train_step(P, [(img, lab, d)], R.RefState(training=True))
    check(got_losses, [losses["loss"], losses["aux_loss"], losses["kt_loss"]], name="512 losses")
    bad = []
    for n, g_ in got_grads.items():
        ref = grads[n]
        if ref is None:
            continue
        # tolerances as in test_mdvit_vs_oracle_128: fp32 GEMMs 3e-3 (mask flips only); bf16x3 perturbs the forward by ~5e-6 (a few more
        # flips): 1e-2, and the bridge -- one image: its BatchNorms normalise over 256 samples, one flipped ReLU moves a channel's
        # gradient by ~1/256 of it, several flip -- is bounded separately (test_gradients_away_from_activation_kinks_match_tightly
        # shows the allowance covers nothing but flips)
        l2_tol, max_tol = (3e-3, 6e-2) if gemm_precision == "fp32" else ((5e-2, 0.5) if n.startswith("bridge.") else (1e-2, 6e-2))
        try:
            check_grad(g_, ref, name=n, l2_tol=l2_tol, max_tol=max_tol)
        except AssertionError as exc:
            bad.append(str(exc))
    assert not bad, f"{len(bad)} gradient tensors off at 512x512: {bad[:6]}"


def _bench_step(m, batches, fuse, side, poison_env=False):
    from mdvit_amd import ops
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.train import mdvit_train_step
    acc = GradAccumulator(m.parameters(), late=[p for n, p in m.named_parameters() if "domain_layer" in n])
    acc.attach_sinks()
    ops.enable_side_stream(side)
    try:
        res = mdvit_train_step(m, batches, optimizer=None, accumulator=acc, merged_sweeps=True, fuse_domains=fuse)
        ops.join_side_stream()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    finally:
        ops.enable_side_stream(False)
        ops.set_grad_sinks(None)
    return {k: float(v) for k, v in res.items()}, grads


@pytest.mark.parametrize("decoder_name", ["MLPFM", "MLP", "DeepLabV3", "Transformer"])
def test_aux_sweep_graph_holds_nothing_the_autograd_engine_would_launch(decoder_name):
    """The data-gradient-only aux sweep may run on a stream of its own only if every kernel of it is launched by this package's Functions: ops.audit_sweep_graph
    finds no torch-native node with a kernel in its backward and no tensor with several consumers outside ops.fork in the aux graph of every peer-head family
    (DeepLabV3's ASPP input -- five consumers -- was one until round 4: adapter gradients ~100 % off in one of eight cold two-stream steps); the audit itself
    reports a planted two-consumer tensor and a planted torch op; and train._aux_graph_is_ours caches its verdict on the model."""
    from mdvit_amd import ops, train
    from mdvit_amd.synthetic import make_step_batches
    batches = make_step_batches(2, 64, rank=0, step=0, device=dev())
    img, lab, sid, dl, G = train._fuse_batches(batches, 4, 4, True)[0]
    Bd = img.shape[0] // G
    m = build_mdvit(23, 64, decoder_name=decoder_name).train()
    out, aux = m(img, dl, [str(int(sid[g * Bd])) for g in range(G)])
    l, la, lk = ops.seg_losses_groups(out, aux, lab, G)
    assert ops.audit_sweep_graph(la) == ([], []), ops.audit_sweep_graph(la)
    assert train._aux_graph_is_ours(m, la) and m._aux_sweep_graph_ok is True
    # planted: a tensor with two consumers (autograd adds the two gradients itself), a torch op with a kernel in its backward
    x = torch.randn(4, 8, 8, 16, device=dev(), requires_grad=True)
    y = ops.upsample_bilinear(x, 16, 16)
    native, fanin = ops.audit_sweep_graph((ops.global_avg_pool(y).sum() + ops.global_avg_pool(y).mean()))
    assert fanin and any("Sum" in n or "Mean" in n for n in native), (native, fanin)


@pytest.mark.parametrize("two_stream", [False, True], ids=["merged_sweeps_on_main", "aux_sweep_on_its_stream"])
def test_bench_step_gradients_are_reproducible_run_to_run(two_stream):
    """The exact bench step (4 domains x bs=4, 512x512, fused forward, weight gradients on the side stream into the bucket sinks) four times from the same seed:
    every gradient tensor agrees to 1e-5 relative L2 between runs (what legitimately differs is the arrival order of LDS float atomics, ~4e-7).  Round 4: all
    46 tensors below the last stage-0 block's MLP were 1e-3 .. 6e-3 off in about every second run -- an LDS-DMA write overtaking another wave's queued ds_read in
    the weight rings of mlp_rc.hip while LDS-atomic kernels of the side stream shared the CU (RC_BARRIER there; tools/probe/step_determinism.py)."""
    from mdvit_amd import train
    from mdvit_amd.synthetic import make_step_batches
    batches = make_step_batches(4, 512, rank=0, step=0, device=dev())
    old = train._two_stream_sweeps
    train._two_stream_sweeps = two_stream
    try:
        res = []
        for _ in range(4):
            m = build_mdvit(23, 512).train()
            res.append(_bench_step(m, batches, 4, True))
            del m
            torch.cuda.empty_cache()
    finally:
        train._two_stream_sweeps = old
    l0, g0 = res[0]
    worst = []
    for l, g in res[1:]:
        for k in ("loss", "aux_loss", "kt_loss"):
            assert abs(l[k] - l0[k]) <= 1e-6 * abs(l0[k]), (k, l[k], l0[k])
        # (a bias in front of a BatchNorm has a zero gradient in exact arithmetic: such tensors hold round-off only and are left out, as in the probe)
        big = max(float(g0[n].double().norm()) for n in g0)
        d = max((float((g[n].double() - g0[n].double()).norm()) / float(g0[n].double().norm()), n) for n in g0 if float(g0[n].double().norm()) > 1e-5 * big)
        worst.append(d)
    assert max(worst)[0] <= 1e-5, f"gradients differ between identical runs: {worst}"


def _repeat_step(make_model, batches, fuse, repeats=3, drop_seeded=False):
    """`repeats` fresh models from one seed through the bench step; the largest relative L2 difference of any gradient tensor against the first run, and its name"""
    import itertools
    from mdvit_amd import ops
    res = []
    for _ in range(repeats):
        ops._key_counter = itertools.count(5)          # the same dropout keys in every run
        torch.manual_seed(1234)                         # the same DropPath draws
        m = make_model()
        res.append(_bench_step(m, batches, fuse, True))
        del m
        torch.cuda.empty_cache()
    l0, g0 = res[0]
    big = max(float(g0[n].double().norm()) for n in g0)
    worst = (0.0, "")
    for l, g in res[1:]:
        for k in ("loss", "aux_loss", "kt_loss"):
            assert abs(l[k] - l0[k]) <= 1e-6 * abs(l0[k]), (k, l[k], l0[k])
        # (a bias in front of a BatchNorm has a zero gradient in exact arithmetic: such tensors hold round-off only and are left out, as in the probe)
        worst = max(worst, max((float((g[n].double() - g0[n].double()).norm()) / float(g0[n].double().norm()), n) for n in g0 if float(g0[n].double().norm()) > 1e-5 * big))
    return worst


@pytest.mark.parametrize("config", ["bs32", "bf16_mode", "dropout", "MDViT_DSN", "MLP", "DeepLabV3", "Transformer"])
def test_step_gradients_are_reproducible_in_the_other_configurations(config):
    """the run-to-run guard of test_bench_step_gradients_are_reproducible_run_to_run where the risk is -- every configuration
    tools/probe/step_determinism.py checked once by hand in round 4, now in the suite: the 128-image fused step (configs[2] per GPU; skipped below 200 GB of device
    memory), the bf16 mode (its own GEMM kernels and bf16 saved tensors), the dropout / DropPath kernel variants (masks re-keyed identically per run), MDViT_DSN and
    the MLP / DeepLabV3 / Transformer peer heads -- each three times from one seed with the aux sweep on its own stream and the weight gradients on the side stream,
    every gradient tensor within 1e-5 relative L2 of the first run.  Every new asynchronous structure (LDS-DMA ring, stream fork) is a new chance of the class of
    bug round 4 found twice; this is where it would show."""
    import mdvit_amd
    from mdvit_amd import ops, train
    from mdvit_amd.synthetic import make_step_batches
    bs, size, decoder, drop = 4, 512, "MLPFM", 0.0
    if config == "bs32":
        if torch.cuda.get_device_properties(0).total_memory < 200 * 2 ** 30:
            pytest.skip("the 128-image step needs ~150 GB")
        bs = 32
    elif config == "dropout":
        drop = 0.1
    elif config in ("MLP", "DeepLabV3", "Transformer"):
        decoder = config
    batches = make_step_batches(bs, size, rank=0, step=0, device=dev())

    def make_model():
        if config == "MDViT_DSN":
            torch.manual_seed(7)                        # MDViT_DSN: domain-specific norms, random init from one torch seed
            m = mdvit_amd.MDViT_DSN(img_size=size, drop_rate=0.0, drop_path_rate=0.0, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                                    decoder_name="MLPFM").to(dev()).train()
            torch.manual_seed(1234)
            return m
        return build_mdvit(23, size, drop=drop, decoder_name=decoder).train()

    old_two, old_prec = train._two_stream_sweeps, ops.gemm_precision()
    train._two_stream_sweeps = True
    if config == "bf16_mode":
        ops.set_gemm_precision("bf16")
    try:
        worst = _repeat_step(make_model, batches, 4)
    finally:
        train._two_stream_sweeps = old_two
        ops.set_gemm_precision(old_prec)
    assert worst[0] <= 1e-5, f"{config}: gradients differ between identical runs: {worst}"


def test_bench_step_fused_forward_equals_per_domain_at_512():
    """The exact bench step (4 domains x bs=4, 512x512, ONE 16-image domain-batched forward, merged sweeps, weight gradients on the
    side stream straight into the bucket sinks) == four per-domain forwards with the same weights: the three losses and every
    gradient tensor; everything finite.  (Parity of the per-domain path with the oracle: the tests above.)"""
    from mdvit_amd.synthetic import make_step_batches
    batches = make_step_batches(4, 512, rank=0, step=0, device=dev())
    res = []
    for fuse, side in ((4, True), (1, False)):
        m = build_mdvit(23, 512).train()
        res.append(_bench_step(m, batches, fuse, side))
        del m
        torch.cuda.empty_cache()
    (la, ga), (lb, gb) = res
    for k in ("loss", "aux_loss", "kt_loss"):
        assert np.isfinite(la[k]) and abs(la[k] - lb[k]) <= 1e-4 * abs(lb[k]), (k, la[k], lb[k])
    bad = []
    for n in ga:
        assert torch.isfinite(ga[n]).all(), n
        try:
            check_grad(ga[n], gb[n], name=n, l2_tol=3e-3, max_tol=6e-2)
        except AssertionError as exc:
            bad.append(str(exc))
    assert not bad, f"{len(bad)} gradient tensors differ between the fused and the per-domain step: {bad[:6]}"


def test_aux_sweep_on_its_own_stream_equals_single_stream_step():
    """The bench step with the data-gradient-only aux sweep on a stream of its own (ops.set_sweep_stream, ops.fork, the gradient stop at
    the first adapter) == the same step with both sweeps on the main stream: same kernels in the same per-stream order, so the losses are
    identical and every gradient agrees to the last bits (window-weight gradients use LDS float atomics).  Repeated: a missing
    dependency between the streams would show up as run-to-run differences."""
    from mdvit_amd import train
    from mdvit_amd.synthetic import make_step_batches
    batches = make_step_batches(2, 256, rank=0, step=0, device=dev())
    prev = train._two_stream_sweeps
    res = []
    try:
        for two in (False, True, True, True):
            train._two_stream_sweeps = two
            m = build_mdvit(31, 256).train()
            res.append(_bench_step(m, batches, 4, True))
            del m
    finally:
        train._two_stream_sweeps = prev
    (l0, g0) = res[0]
    for (l1, g1) in res[1:]:
        for k in ("loss", "aux_loss", "kt_loss"):
            assert l1[k] == l0[k], (k, l0[k], l1[k])
        for n in g0:
            assert torch.isfinite(g1[n]).all(), n
            d = float((g1[n].double() - g0[n].double()).norm()) / max(float(g0[n].double().norm()), 1e-30)
            assert d <= 1e-5, f"{n}: two-stream vs single-stream gradients differ by {d:.2e}"


def test_side_stream_hold_bound_releases_in_stream_order_with_equal_results():
    """ops._side_hold_limit: past the bound the stream that owns a block's saved tensors WAITS (on the GPU) for the weight-gradient stream to pass that
    block and the tensors go back to the pool in stream order, instead of staying allocated until a completed event is seen (bs=32: peak allocation
    129.9 -> 86.9 GiB).  With a bound of ONE byte every block takes that route, and the very next allocations reuse the memory the weight-gradient
    kernels were reading: the step == the unbounded step (losses identical, gradients to the last bits), three times over -- a wait on the wrong
    stream or event would show as garbage or as run-to-run differences."""
    from mdvit_amd import ops
    from mdvit_amd.synthetic import make_step_batches
    batches = make_step_batches(2, 256, rank=0, step=0, device=dev())
    prev = ops._side_hold_limit
    res = []
    try:
        for limit in (0, 1, 1, 1):
            ops._side_hold_limit = limit
            m = build_mdvit(37, 256).train()
            res.append(_bench_step(m, batches, 4, True))
            assert not ops._side_groups and ops._side_held[0] == 0          # the join released every group
            del m
    finally:
        ops._side_hold_limit = prev
    (l0, g0) = res[0]
    for (l1, g1) in res[1:]:
        for k in ("loss", "aux_loss", "kt_loss"):
            assert l1[k] == l0[k], (k, l0[k], l1[k])
        for n in g0:
            assert torch.isfinite(g1[n]).all(), n
            d = float((g1[n].double() - g0[n].double()).norm()) / max(float(g0[n].double().norm()), 1e-30)
            assert d <= 1e-5, f"{n}: bounded-hold vs unbounded gradients differ by {d:.2e}"


def test_aux_sweep_runs_next_to_the_full_sweep_not_behind_it():
    """The GPU has four hardware queues.  With a fifth stream in the process the aux sweep's stream shares the main stream's queue and the whole
    data-gradient-only sweep runs BEHIND the full sweep (it cost 8 % of the step for half a round).  Guard: in the bench step the aux sweep's stream
    becomes runnable right after the forward -- long before the full sweep's last kernel.  (Skipped when the host is the limit of the step on
    this box: the aux sweep is then enqueued late whatever the queues do.)"""
    import time
    import mdvit_amd
    from mdvit_amd import ops, train
    from mdvit_amd.optim import FusedAdamW
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.synthetic import make_step_batches
    if ops.gemm_precision() != "bf16x3":
        pytest.skip("timing property of the default (bf16x3) step")
    torch.manual_seed(0)
    model = mdvit_amd.MDViT(img_size=512, drop_rate=0.1, drop_path_rate=0.1, conv_norm=torch.nn.BatchNorm2d, adapt_method="Sup", num_domains=4,
                            decoder_name="MLPFM").to(dev()).train()
    ops.enable_side_stream(True)
    accum = GradAccumulator(model.parameters(), late=[p for n, p in model.named_parameters() if "domain_layer" in n])
    accum.attach_sinks()
    opt = FusedAdamW(accum, lr=1e-4, weight_decay=0.05)
    pool = [make_step_batches(4, 512, rank=0, step=s_, device=dev()) for s_ in range(2)]
    try:
        def step(i, evs=None):
            return train.mdvit_train_step(model, pool[i % 2], optimizer=opt, accumulator=accum, merged_sweeps=True, fuse_domains=4, phase_events=evs)
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            step(0)
            train._timeline = []
            evs = []
            h0 = time.perf_counter()
            step(1, evs)
            host_ms = 1e3 * (time.perf_counter() - h0)
            tl, train._timeline = train._timeline, None
            torch.cuda.synchronize()
            e0 = evs[0][1]
            t = {tag: e0.elapsed_time(e) for tag, e, _ in tl}
            t.update({tag: e0.elapsed_time(e) for tag, e in evs[1:]})
            best = (host_ms, t)
    finally:
        train._timeline = None
        ops.set_grad_sinks(None); ops.enable_side_stream(False)
    host_ms, t = best
    fwd, runnable, main_last, opt_t = t["fwd"], t["aux sweep: runnable (its stream)"], t["sweep on main: last kernel"], t["opt"]
    if host_ms > 0.8 * opt_t:
        pytest.skip(f"host-bound on this box (enqueue {host_ms:.1f} ms of a {opt_t:.1f} ms step)")
    assert runnable < fwd + 0.25 * (main_last - fwd), \
        f"the aux sweep's stream became runnable at {runnable:.1f} ms: forward ends at {fwd:.1f}, the full sweep at {main_last:.1f} -- a stream too many?"


def test_peer_heads_on_their_own_streams_equal_heads_on_the_main_stream():
    """MDVIT_PEER_STREAMS=1 (opt-in since the aux sweep has its own stream): the four peer heads of a domain-batched forward, and
    their backward chains, on a stream each == the default (heads on the main stream) -- identical losses, gradients to the last bits;
    repeated so that a missing cross-stream dependency shows up as a run-to-run difference."""
    from mdvit_amd import ops
    from mdvit_amd.synthetic import make_step_batches
    batches = make_step_batches(2, 256, rank=0, step=0, device=dev())
    prev = ops._use_peer_streams
    res = []
    try:
        for peers in (False, True, True):
            ops._use_peer_streams = peers
            m = build_mdvit(31, 256).train()
            res.append(_bench_step(m, batches, 4, True))
            del m
    finally:
        ops._use_peer_streams = prev
    (l0, g0) = res[0]
    for (l1, g1) in res[1:]:
        for k in ("loss", "aux_loss", "kt_loss"):
            assert l1[k] == l0[k], (k, l0[k], l1[k])
        for n in g0:
            assert torch.isfinite(g1[n]).all(), n
            d = float((g1[n].double() - g0[n].double()).norm()) / max(float(g0[n].double().norm()), 1e-30)
            assert d <= 1e-5, f"{n}: peer-stream vs main-stream gradients differ by {d:.2e}"


def test_bs32_shape_fused_128_image_forward_matches_per_domain_forwards():
    """BASELINE configs[2]'s per-GPU shape: one 128-image (4 domains x 32) domain-batched train-mode forward at 512x512 -- tensors
    beyond 4 GiB -- against four 32-image per-domain forwards with the same weights, on a strided sample of the logits."""
    from mdvit_amd import ops
    from mdvit_amd.synthetic import make_domain_batch
    B = 32
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * (1 << 30):
        pytest.skip("needs ~80 GB of free HBM")
    m = build_mdvit(29, 512).train()
    imgs = [make_domain_batch(B, 512, d, 4321, dev())[0] for d in range(4)]
    dls = [F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev()) for d in range(4)]
    with torch.no_grad():
        per = []
        for d in range(4):
            o, a = m(imgs[d], dls[d], str(d))
            per.append((o[:, :, ::37, ::41].clone(), a[:, :, ::37, ::41].clone()))
        o, a = m(torch.cat(imgs, 0), torch.cat(dls, 0), ["0", "1", "2", "3"])       # the model keeps BatchNorm statistics per domain batch
        assert o.shape == (4 * B, 1, 512, 512) and torch.isfinite(o).all() and torch.isfinite(a).all()
        for d in range(4):
            check(o[d * B:(d + 1) * B, :, ::37, ::41], per[d][0], tol=1e-4, name=f"fused out, domain {d}")
            check(a[d * B:(d + 1) * B, :, ::37, ::41], per[d][1], tol=1e-4, name=f"fused aux, domain {d}")


def test_bs32_full_step_fused_128_images_equals_four_per_domain_steps():
    """BASELINE configs[2]'s per-GPU step, exactly as bench.py --batch 32 runs it: ONE 128-image domain-batched forward at 512x512, merged
    sweeps (full sweep + data-gradient-only aux sweep on its own stream), weight gradients on the side stream into the bucket sinks, the
    adapters in late buckets -- split-K weight gradients over 2 M tokens, tensors beyond 4 GiB in every backward kernel -- against four
    32-image per-domain forwards + sweeps with the same weights: the three summed losses and EVERY gradient tensor."""
    from mdvit_amd.synthetic import make_step_batches
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~200 GB of free HBM")
    batches = make_step_batches(32, 512, rank=0, step=0, device=dev())
    res = []
    for fuse, side in ((4, True), (1, False)):
        m = build_mdvit(37, 512).train()
        res.append(_bench_step(m, batches, fuse, side))
        del m
        torch.cuda.empty_cache()
    (la, ga), (lb, gb) = res
    for k in ("loss", "aux_loss", "kt_loss"):
        assert np.isfinite(la[k]) and abs(la[k] - lb[k]) <= 1e-4 * abs(lb[k]), (k, la[k], lb[k])
    bad = []
    for n in ga:
        assert torch.isfinite(ga[n]).all(), n
        try:
            check_grad(ga[n], gb[n], name=n, l2_tol=3e-3, max_tol=6e-2)
        except AssertionError as exc:
            bad.append(str(exc))
    assert not bad, f"{len(bad)} gradient tensors differ between the fused 128-image step and the per-domain steps: {bad[:6]}"


def test_bs16_step_in_the_bf16_mode_at_512_is_finite_and_within_its_drift_bound():
    """BASELINE configs[3] ("bs=16, mixed bf16 / fp32 loss"): the 64-image fused step at 512x512 with the GEMMs in the bf16 speed mode (one
    bf16 plane per operand; norms, softmax, losses and storage fp32) against the same step in the parity arithmetic (bf16x3): everything
    finite, the three losses within 2e-2, the logits of the first images within 5e-2 -- the bound test_bf16_speed_mode_drift... holds at 64x64."""
    from mdvit_amd import ops
    from mdvit_amd.synthetic import make_step_batches
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs ~120 GB of free HBM")
    batches = make_step_batches(16, 512, rank=0, step=0, device=dev())
    prev = ops.gemm_precision()
    res = {}
    try:
        for mode in ("bf16x3", "bf16"):
            ops.set_gemm_precision(mode)
            m = build_mdvit(41, 512).train()
            losses, grads = _bench_step(m, batches, 4, True)
            with torch.no_grad():
                m.eval()
                o, a = m(batches[1][0][:2], F.one_hot(torch.full((2,), 1, dtype=torch.long), 4).float().to(dev()), "1")
            res[mode] = (losses, grads, o.clone(), a.clone())
            del m
            torch.cuda.empty_cache()
    finally:
        ops.set_gemm_precision(prev)
    lx, gx, ox, ax = res["bf16x3"]
    lb, gb, ob, ab = res["bf16"]
    assert all(np.isfinite(v) for v in lb.values()) and all(torch.isfinite(g).all() for g in gb.values())
    e_loss = max(abs(lb[k] - lx[k]) / abs(lx[k]) for k in ("loss", "aux_loss", "kt_loss"))
    e_out, e_aux = relerr(ob, ox), relerr(ab, ax)
    worst = max(float((gb[n].double() - gx[n].double()).norm() / max(float(gx[n].double().norm()), 1e-30)) for n in gx if float(gx[n].abs().max()) > 1e-7)
    print(f"bf16 speed mode at bs=16, 512x512 vs bf16x3: losses {e_loss:.2e}, logits {e_out:.2e} / {e_aux:.2e}, worst gradient tensor rel L2 {worst:.2e}")
    assert e_loss <= 2e-2 and e_out <= 5e-2 and e_aux <= 5e-2, (e_loss, e_out, e_aux)
    assert e_out > 1e-5


def test_gradients_away_from_activation_kinks_match_tightly():
    """Why the whole-model gradient tolerances are 1e-2 / 6 %: the residual error IS mask flips at the ReLU / Hardswish kinks.
    The oracle reports, per BatchNorm output, the distance to the activation's kink; with inputs whose smallest distance is far above
    fp32 round-off no derivative can flip, and then every gradient tensor must agree at 2e-4 relative L2 (fp32 GEMMs) -- the
    allowance is not hiding an error of any other kind."""
    from mdvit_amd import ops
    from mdvit_amd.losses import domain_losses
    from oracle import mdvit_ref as R
    from oracle.gen_golden import synth_image, synth_label
    from oracle.params import make_params
    S, B, d = 64, 2, 1
    prev = ops.gemm_precision()
    ops.set_gemm_precision("fp32")
    try:
        best = None
        for seed in range(40, 52):                       # pick the input whose closest BatchNorm output is farthest from a kink
            pn = make_params(seed, model="MDViT", adapt_method="Sup")
            img, lab = synth_image(2000 + seed, B, S, S), synth_label(2100 + seed, B, S, S)
            margin = R.kink_margin(R.to_torch(pn), [(img, lab, d)], R.RefState(training=True))
            if best is None or margin > best[0]:
                best = (margin, seed, pn, img, lab)
        margin, seed, pn, img, lab = best
        assert margin > 2e-6, f"no test input with a kink margin above round-off (best {margin:.2e})"
        losses, grads = R.mdvit_train_step(R.to_torch(pn), [(img, lab, d)], R.RefState(training=True))
        m = build_mdvit(seed, S).train()
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
        out, aux = m(img.to(dev()), dl, str(d))
        l, la, lk = domain_losses(out, aux, lab.to(dev()))
        da = [p for n, p in m.named_parameters() if "domain_layer" in n]
        for p in da:
            p.requires_grad = False
        la.backward(retain_graph=True)
        for p in da:
            p.requires_grad = True
        (0.5 * lk + 0.5 * l).backward()
    finally:
        ops.set_gemm_precision(prev)
    bad = []
    for n, p in m.named_parameters():
        ref = grads[n]
        if ref is None:
            continue
        try:
            check_grad(p.grad, ref, name=n, l2_tol=2e-4, max_tol=2e-3)
        except AssertionError as exc:
            bad.append(str(exc))
    assert not bad, f"kink margin {margin:.2e}: {len(bad)} gradient tensors off: {bad[:6]}"


def test_bf16_speed_mode_drift_is_bounded_and_reported():
    """The bf16 speed mode (GEMM operands one bf16 plane) is NOT the parity mode: its drift against the oracle is measured here and
    held to 5e-2 on logits / 2e-2 on the losses (bf16 autocast of the reference itself drifts 2-3 % on logits)."""
    from mdvit_amd import ops
    from mdvit_amd.losses import domain_losses
    from oracle import mdvit_ref as R
    from oracle.gen_golden import synth_image, synth_label
    from oracle.params import make_params
    S, B, d = 64, 2, 2
    pn = make_params(31, model="MDViT", adapt_method="Sup")
    img, lab = synth_image(3100, B, S, S), synth_label(3101, B, S, S)
    P = R.to_torch(pn)
    with torch.no_grad():
        ro, ra = R.mdvit_forward({k: v.clone() for k, v in P.items()}, img, F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float(), str(d),
                                 R.RefState(training=True))
    losses, _ = R.mdvit_train_step(P, [(img, lab, d)], R.RefState(training=True))
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    try:
        m = build_mdvit(31, S).train()
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
        out, aux = m(img.to(dev()), dl, str(d))
        l, la, lk = domain_losses(out, aux, lab.to(dev()))
        (l + la + lk).backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    finally:
        ops.set_gemm_precision(prev)
    e_out, e_aux = relerr(out, ro), relerr(aux, ra)
    e_loss = relerr(torch.stack([l, la, lk]), [losses["loss"], losses["aux_loss"], losses["kt_loss"]])
    print(f"bf16 speed mode drift vs oracle: logits {e_out:.2e}, aux logits {e_aux:.2e}, losses {e_loss:.2e}")
    assert e_out <= 5e-2 and e_aux <= 5e-2 and e_loss <= 2e-2, (e_out, e_aux, e_loss)
    assert e_out > 1e-5          # it IS a different arithmetic: if this were parity-class the mode would not be running
