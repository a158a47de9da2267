"""TransFuse_S_adapt path (BASELINE configs[4]) on the GPU: every kernel of csrc/transfuse.hip (through the C ABI) against a plain
torch reference, the module surface (630 state_dict keys), and the whole model -- forward, structure losses and the gradients of
the train step -- against the fixture the reference's own TransFuse_S_adapt produced and against the CPU oracle."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def check(a, b, tol=1e-4, name=""):
    assert tuple(a.shape) == tuple(b.shape), f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = relerr(a, b)
    assert math.isfinite(e) and e <= tol, f"{name}: rel-to-max error {e:.3e} > {tol}"


def grads_of(fn, inputs, gout):
    ins = [t.clone().requires_grad_(True) for t in inputs]
    out = fn(*ins)
    out.backward(gout.to(out.device))
    return out, [t.grad for t in ins]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def test_imgconv7_and_maxpool():
    from mdvit_amd import transfuse as T
    img, w = rnd(2, 3, 40, 56, seed=1), rnd(64, 3, 7, 7, seed=2, scale=0.1)
    g = rnd(2, 20, 28, 64, seed=3)
    ref, gr = grads_of(lambda w: nhwc(F.conv2d(img.double(), w.double(), None, 2, 3)), [w], g.double())
    out, go = grads_of(lambda w: T._ImgConv.apply(img.to(dev()), w), [w.to(dev())], g)
    check(out, ref, name="conv1")
    check(go[0], gr[0], name="conv1 dw")
    out, go = grads_of(lambda w: T.stem_conv7(img.to(dev()), w), [w.to(dev())], g)          # the product path: im2col + GEMMs
    check(out, ref, name="conv1 (im2col + GEMM)")
    check(go[0], gr[0], name="conv1 dw (im2col + GEMM)")
    x = rnd(2, 64, 21, 30, seed=4)
    g2 = rnd(2, 11, 15, 64, seed=5)
    ref, gr = grads_of(lambda x: nhwc(F.max_pool2d(x.double(), 3, 2, 1)), [x], g2.double())
    out, go = grads_of(lambda x: T._MaxPool.apply(x), [nhwc(x).to(dev())], g2)
    check(out, ref, name="maxpool")
    check(nchw(go[0]), gr[0], name="maxpool dx")


@pytest.mark.parametrize("B,H,W,C,scale", [(2, 16, 16, 8, 2), (1, 5, 7, 3, 4), (2, 16, 16, 1, 16)])
def test_resize_align_corners(B, H, W, C, scale):
    from mdvit_amd import transfuse as T
    x, g = rnd(B, C, H, W, seed=6), rnd(B, H * scale, W * scale, C, seed=7)
    ref, gr = grads_of(lambda x: nhwc(F.interpolate(x.double(), scale_factor=scale, mode="bilinear", align_corners=True)), [x], g.double())
    out, go = grads_of(lambda x: T.resize_ac(x, scale), [nhwc(x).to(dev())], g)
    check(out, ref, name="resize")
    check(nchw(go[0]), gr[0], name="resize dx")


def test_elementwise_gates_pool():
    from mdvit_amd import transfuse as T
    a, b, g = rnd(3, 9, 11, 16, seed=8), rnd(3, 9, 11, 16, seed=9), rnd(3, 9, 11, 16, seed=10)
    ref, gr = grads_of(lambda a, b: F.relu(a.double() + b.double()), [a, b], g.double())
    out, go = grads_of(lambda a, b: T.add_relu(a, b), [a.to(dev()), b.to(dev())], g)
    check(out, ref, name="add_relu"); check(go[0], gr[0], name="add_relu da"); check(go[1], gr[1], name="add_relu db")
    ref, gr = grads_of(lambda a, b: a.double() * b.double(), [a, b], g.double())
    out, go = grads_of(lambda a, b: T._Mul.apply(a, b), [a.to(dev()), b.to(dev())], g)
    check(out, ref, name="mul"); check(go[0], gr[0], name="mul da"); check(go[1], gr[1], name="mul db")
    s0, s1 = rnd(3, 99, seed=11, scale=2.0), rnd(3, 16, seed=12, scale=2.0)
    ref, gr = grads_of(lambda x, s: torch.sigmoid(s.double()).view(3, 9, 11, 1) * x.double(), [a, s0], g.double())
    out, go = grads_of(lambda x, s: T._Gate.apply(x, s, 0), [a.to(dev()), s0.to(dev())], g)
    check(out, ref, name="spatial gate"); check(go[0], gr[0], name="spatial gate dx"); check(go[1], gr[1], name="spatial gate ds")
    ref, gr = grads_of(lambda x, s: torch.sigmoid(s.double()).view(3, 1, 1, 16) * x.double(), [a, s1], g.double())
    out, go = grads_of(lambda x, s: T._Gate.apply(x, s, 1), [a.to(dev()), s1.to(dev())], g)
    check(out, ref, name="channel gate"); check(go[0], gr[0], name="channel gate dx"); check(go[1], gr[1], name="channel gate ds")
    x = rnd(2, 7, 9, 40, seed=13); g2 = rnd(2, 7, 9, 2, seed=14)
    ref, gr = grads_of(lambda x: torch.stack((x.double().max(-1)[0], x.double().mean(-1)), -1), [x], g2.double())
    out, go = grads_of(lambda x: T._ChanPool.apply(x), [x.to(dev())], g2)
    check(out, ref, name="channel pool"); check(go[0], gr[0], name="channel pool dx")
    x = rnd(2, 13, 18, 32, seed=15); g3 = rnd(2, 7, 9, 32, seed=16)
    ref, gr = grads_of(lambda x: x.double()[:, ::2, ::2, :], [x], g3.double())
    out, go = grads_of(lambda x: T._Subsample2.apply(x), [x.to(dev())], g3)
    check(out, ref, name="subsample"); check(go[0], gr[0], name="subsample dx")


def test_spatial_conv7_and_single_channel_batchnorm():
    from mdvit_amd import transfuse as T
    x, w, g = rnd(2, 2, 12, 17, seed=17), rnd(1, 2, 7, 7, seed=18, scale=0.2), rnd(2, 12, 17, seed=19)
    ref, gr = grads_of(lambda x, w: F.conv2d(x.double(), w.double(), None, 1, 3)[:, 0], [x, w], g.double())
    out, go = grads_of(lambda x, w: T._Conv7x7_2to1.apply(x, w), [nhwc(x).to(dev()), w.to(dev())], g)
    check(out, ref, name="conv7"); check(nchw(go[0]), gr[0], name="conv7 dx"); check(go[1], gr[1], name="conv7 dw")
    for training in (True, False):
        x = rnd(3, 10, 14, seed=20, scale=2.0) + 0.3
        ga, be = torch.tensor([1.3]), torch.tensor([-0.2])
        rm, rv = torch.tensor([0.1]), torch.tensor([1.7])
        bn = torch.nn.BatchNorm2d(1).double()
        with torch.no_grad():
            bn.weight.copy_(ga); bn.bias.copy_(be); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        bn.train(training)
        xr = x.clone().double().requires_grad_(True)
        yr = bn(xr.view(3, 1, 10, 14)).view(3, 10, 14)
        gg = rnd(3, 10, 14, seed=21)
        yr.backward(gg.double())
        m = T.BatchNorm1ch().to(dev())
        with torch.no_grad():
            m.weight.copy_(ga); m.bias.copy_(be); m.running_mean.copy_(rm); m.running_var.copy_(rv)
        m.train(training)
        xg = x.to(dev()).requires_grad_(True)
        y = m(xg)
        y.backward(gg.to(dev()))
        check(y, yr, name=f"bn1 y train={training}"); check(xg.grad, xr.grad, name="bn1 dx")
        check(m.weight.grad, bn.weight.grad, name="bn1 dgamma"); check(m.bias.grad, bn.bias.grad, name="bn1 dbeta")
        check(m.running_mean, bn.running_mean, name="bn1 running_mean"); check(m.running_var, bn.running_var, name="bn1 running_var")
        assert int(m.num_batches_tracked) == int(bn.num_batches_tracked)


@pytest.mark.parametrize("N", [256, 128])
def test_softmax_attention_with_domain_adapter(N):
    """Attention_Sup core (vision_transformer.py:148-169): softmax(q k^T / 8) v scaled by the head-softmax adapter, fwd + all grads.
    N = 256 (the DeiT trunk's shape) runs on the fp32-MFMA kernels of csrc/sdpa.hip, other lengths on the LDS-tiled VALU kernels."""
    from mdvit_amd import transfuse as T
    B, heads, D = 2, 6, 64
    Cn, hid = heads * D, 192
    qkv = rnd(B, N, 3 * Cn, seed=22)
    label = F.one_hot(torch.tensor([1, 3]), 4).float()
    W1, b1 = rnd(hid, 4, seed=23), rnd(hid, seed=24, scale=0.1)
    W2, b2 = rnd(Cn, hid, seed=25, scale=0.2), rnd(Cn, seed=26, scale=0.1)
    g = rnd(B, N, Cn, seed=27)

    def ref_fn(qkv, W1, b1, W2, b2):
        q, k, v = qkv.double().reshape(B, N, 3, heads, D).permute(2, 0, 3, 1, 4)
        attn = ((q @ k.transpose(-2, -1)) * D ** -0.5).softmax(-1)
        o = attn @ v
        da = F.linear(F.relu(F.linear(label.double(), W1.double(), b1.double())), W2.double(), b2.double())
        da = torch.softmax(da.view(B, heads, 1, D), dim=1)
        return (da * o).transpose(1, 2).reshape(B, N, Cn)
    ref, gr = grads_of(ref_fn, [qkv, W1, b1, W2, b2], g.double())
    out, go = grads_of(lambda qkv, W1, b1, W2, b2: T._SDPA.apply(qkv, label.to(dev()), W1, b1, W2, b2, heads),
                       [t.to(dev()) for t in (qkv, W1, b1, W2, b2)], g)
    check(out, ref, name="sdpa out")
    for n, a, r in zip(("dqkv", "dW1", "db1", "dW2", "db2"), go, gr):
        check(a, r, tol=2e-4, name="sdpa " + n)


def test_structure_loss_and_patch_embed_pieces():
    from mdvit_amd import transfuse as T
    from oracle import transfuse_ref as R
    from oracle.gen_golden import synth_label
    B, S = 3, 64
    mask = synth_label(77, B, S, S)
    pred = rnd(B, 1, S, S, seed=28, scale=4.0)
    pr = pred.clone().double().requires_grad_(True)
    lr = R.structure_loss(pr, mask.double())
    lr.backward()
    pg = pred.to(dev()).requires_grad_(True)
    lg = T.structure_loss(pg, mask.to(dev()))
    (2.5 * lg).backward()
    assert abs(float(lg) - float(lr)) <= 1e-5 * abs(float(lr))
    check(pg.grad, 2.5 * pr.grad, tol=2e-4, name="structure_loss dpred")
    weit = T.structure_weight(mask.to(dev()))
    check(weit, 1 + 5 * torch.abs(F.avg_pool2d(mask, 31, 1, 15) - mask), tol=1e-5, name="weit")
    # PatchEmbed = gather + Linear; positional embedding add
    img = rnd(2, 3, 64, 48, seed=29)
    w, b = rnd(32, 3, 16, 16, seed=30, scale=0.05), rnd(32, seed=31)
    ref = F.conv2d(img.double(), w.double(), b.double(), 16).flatten(2).transpose(1, 2)
    from mdvit_amd import ops
    from mdvit_amd._lib import call
    patches = torch.empty((2 * 4 * 3, 768), device=dev())
    call("mdvit_patchify", ops._p(img.to(dev())), ops._p(patches), 2, 3, 64, 48, 16, ops._stream())
    out = ops.linear(patches, w.to(dev()).view(32, -1), b.to(dev())).view(2, 12, 32)
    check(out, ref, name="patch embed")
    x, pe, g = rnd(3, 12, 32, seed=32), rnd(1, 12, 32, seed=33), rnd(3, 12, 32, seed=34)
    ref, gr = grads_of(lambda x, pe: x.double() + pe.double(), [x, pe], g.double())
    o2, go = grads_of(lambda x, pe: T._AddPos.apply(x, pe), [x.to(dev()), pe.to(dev())], g)
    check(o2, ref, name="pos add"); check(go[0], gr[0], name="pos dx"); check(go[1], gr[1], name="pos dpe")
    # Dropout2d: whole (sample, channel) planes dropped, survivors scaled, the backward uses the same mask
    x = torch.ones(8, 5, 7, 64, device=dev(), requires_grad=True)
    y = T.dropout2d(x, 0.25, True)
    y.sum().backward()
    per_plane = y.detach().view(8, 35, 64)
    assert bool(((per_plane == 0).all(1) | (per_plane == per_plane[:, :1]).all(1)).all())
    keep = float((per_plane[:, 0, :] != 0).float().mean())
    assert 0.6 < keep < 0.9 and abs(float(per_plane.max()) - 1 / 0.75) < 1e-5
    assert torch.equal(x.grad, y.detach())


def compare_grads(got, ref, precision):
    """Per-tensor relative L2 against the reference gradients.  Measured on this model (two 256x256 images per domain: the deepest
    BatchNorms normalise over 512 samples, 26 ReLU/BatchNorm pairs in the CNN branch alone): fp32 GEMMs -- median 2.7e-3, worst 1e-2
    (4e-2 on the one-element BatchNorm2d(1) weights, each a sum of cancelling terms); bf16x3 GEMMs move the LOGITS by 2.7e-4 (bar: 1e-3)
    and the gradients with them -- median 1.6e-2, worst multi-element tensor 3.9e-2, one-element tensors up to 0.25.  Tensors that are
    analytically zero (conv biases in front of a train-mode BatchNorm) hold round-off only and are skipped."""
    big = max(float(v.double().norm()) for v in ref.values() if v is not None)
    med_tol, tol, tol1 = (6e-3, 2e-2, 8e-2) if precision == "fp32" else (3e-2, 6e-2, 0.4)      # calibrated: test_transfuse_gradient_error_is_the_round_off_class_...
    errs, bad = [], []
    for n, g_ in got.items():
        r = ref.get(n)
        if r is None:
            assert g_ is None or float(g_.abs().max()) == 0.0, n
            continue
        rn = float(r.double().norm())
        if rn <= 1e-5 * big:
            continue
        e = float((g_.double() - r.double()).norm() / rn)
        errs.append(e)
        if not e <= (tol1 if r.numel() == 1 else tol):
            bad.append(f"{n}: rel L2 {e:.2e}")
    assert not bad, f"{len(bad)} gradient tensors off: {bad[:8]}"
    assert len(errs) > 300 and float(np.median(errs)) <= med_tol, f"median gradient error {np.median(errs):.2e} over {len(errs)} tensors"


def _build(seed, drop=0.0):
    from mdvit_amd.transfuse import TransFuse_S_adapt
    from oracle import transfuse_ref as R
    pn = R.make_params(seed)
    m = TransFuse_S_adapt(num_classes=1, drop_rate=drop, normal_init=False, pretrained=False, num_domains=4)
    sd = m.state_dict()
    assert set(sd) == set(pn), (sorted(set(sd) - set(pn))[:5], sorted(set(pn) - set(sd))[:5])
    assert len(sd) == 630 and all(tuple(sd[k].shape) == tuple(np.asarray(v).shape) for k, v in pn.items())
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in pn.items()}, strict=True)
    return m.to(dev()), pn


def test_transfuse_state_dict_surface_and_size_check():
    m, pn = _build(12)
    assert sum(p.numel() for p in m.parameters()) == 26873877          # the reference's 26.87 M parameters
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 512, 512, device=dev()), torch.zeros(1, 4, device=dev()))        # pos_embed holds 16x16 tokens (DeiT.py:134)


def test_transfuse_step_vs_reference_golden(golden, gemm_precision):
    """two domains x 2 images at 256x256: the three logit maps, the step losses and the norm of every gradient tensor against the
    fixture the reference's TransFuse_S_adapt produced; BatchNorm running statistics"""
    from mdvit_amd import transfuse as T
    from oracle.gen_golden import synth_image, synth_label, grad_digest
    g = golden("transfuse_step_256")
    S, B, seed = [int(v) for v in g["meta"]]
    m, _ = _build(seed)
    m.train()
    tot = None
    for d in (1, 3):
        img, lab = synth_image(1200 + d, B, S, S).to(dev()), synth_label(1300 + d, B, S, S).to(dev())
        dl = F.one_hot(torch.full((B,), d, dtype=torch.long), 4).float().to(dev())
        m4, m3, m2 = m(img, dl)
        assert m4.shape == m3.shape == m2.shape == (B, 1, S, S)
        for nm, t in (("map_x", m4), ("map_1", m3), ("map_2", m2)):
            ref = torch.from_numpy(g[f"{nm}_{d}"])
            check(t.detach().reshape(-1)[::61].cpu(), ref, tol=1e-3, name=f"{nm} domain {d}")
        weit = T.structure_weight(lab)
        loss = 0.5 * T.structure_loss(m2, lab, weit) + 0.3 * T.structure_loss(m3, lab, weit) + 0.2 * T.structure_loss(m4, lab, weit)
        assert abs(float(loss) - float(g[f"loss_{d}"])) <= 1e-3 * abs(float(g[f"loss_{d}"]))
        tot = loss if tot is None else tot + loss
    tot.backward()
    names, norms, _ = grad_digest({n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in m.named_parameters()})
    assert list(names) == [str(n) for n in g["grad_names"]]
    ref = g["grad_norms"]
    rel = np.abs(norms - ref) / np.maximum(ref, 1e-6 * ref.max())
    numel = {n: p.numel() for n, p in m.named_parameters()}
    lim = np.array([(8e-2 if gemm_precision == "fp32" else 0.4) if numel[str(n)] == 1 else (2e-2 if gemm_precision == "fp32" else 5e-2) for n in names])
    order = np.argsort(-(rel / lim))[:8]      # (tolerances: see compare_grads)
    assert (rel <= lim).all(), "gradient norms off: " + ", ".join(f"{names[i]} {rel[i]:.2e} (ref {ref[i]:.2e})" for i in order)
    for k in ("resnet.bn1.running_mean", "up_c.residual.bn1.running_var", "up_c_2_2.attn_block.psi.1.running_mean"):
        check(m.state_dict()[k].cpu(), torch.from_numpy(g["buf__" + k]), tol=1e-3, name=k)
    for n in ("resnet.conv1.weight", "transformer.blocks.0.attn.domain_layer.2.weight", "transformer.pos_embed", "up_c.spatial.conv.weight",
              "up_c_1_2.attn_block.psi.0.weight", "final_x.2.conv.weight", "up_c.fc1.weight"):
        got = dict(m.named_parameters())[n].grad.reshape(-1)[::7].cpu()
        refg = torch.from_numpy(g["grad__" + n])
        l2 = float((got.double() - refg.double()).norm() / max(float(refg.double().norm()), 1e-30))
        assert l2 <= (2e-2 if gemm_precision == "fp32" else 8e-2), f"{n}: sampled gradient rel L2 {l2:.2e}"


def test_transfuse_train_step_harness_vs_oracle(gemm_precision):
    """mdvit_amd.transfuse.transfuse_train_step (gradient accumulator + sinks + side stream) == the oracle's step: losses and gradients"""
    from mdvit_amd import ops
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.transfuse import transfuse_train_step
    from oracle import transfuse_ref as R
    from oracle.gen_golden import synth_image, synth_label
    B, S = 2, 256
    m, pn = _build(5)
    m.train()
    batches_cpu = [(synth_image(2200 + d, B, S, S), synth_label(2300 + d, B, S, S), d) for d in (0, 2)]
    acc = GradAccumulator(m.parameters()); acc.attach_sinks(); ops.enable_side_stream(True)
    try:
        res = transfuse_train_step(m, [(i.to(dev()), l.to(dev()), torch.full((B,), d, dtype=torch.long)) for i, l, d in batches_cpu], accumulator=acc)
        torch.cuda.synchronize()
        got = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    finally:
        ops.enable_side_stream(False); ops.set_grad_sinks(None)
    losses, grads = R.transfuse_train_step(R.to_torch(pn), batches_cpu, R.TFState(training=True))
    check(res["per_domain"].cpu(), torch.tensor(losses), tol=1e-3, name="per-domain losses")
    compare_grads(got, grads, gemm_precision)


def test_transfuse_gradient_error_is_the_round_off_class_of_the_reference_arithmetic(gemm_precision):
    """Why compare_grads allows 2e-2 (fp32 GEMMs) / 8e-2 (bf16x3) per tensor.  MDViT has a kink-margin test: an input on which no BatchNorm
    output sits within round-off of its activation's kink, where all gradients agree at 2e-4.  TransFuse offers no such input -- two 256x256
    images push ~2e7 values through ReLU / max-pool / channel-max, the closest one always lies within fp32 round-off of its kink -- so the
    allowance is CALIBRATED instead: the oracle is run in fp64 (the derivative masks of exact arithmetic) and in fp32 (the reference's own
    arithmetic).  The distance of the fp32 oracle from the fp64 one is what two correct implementations differ by; the HIP step must sit in
    the same class: median and 90th percentile of its per-tensor error against fp64 within a small factor of the fp32 oracle's."""
    from mdvit_amd import ops
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.transfuse import transfuse_train_step
    from oracle import transfuse_ref as R
    from oracle.gen_golden import synth_image, synth_label
    B, S = 2, 256
    m, pn = _build(9)
    m.train()
    batches_cpu = [(synth_image(4200 + d, B, S, S), synth_label(4300 + d, B, S, S), d) for d in (1, 2)]
    acc = GradAccumulator(m.parameters()); acc.attach_sinks(); ops.enable_side_stream(True)
    try:
        transfuse_train_step(m, [(i.to(dev()), l.to(dev()), torch.full((B,), d, dtype=torch.long)) for i, l, d in batches_cpu], accumulator=acc)
        torch.cuda.synchronize()
        got = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}
    finally:
        ops.enable_side_stream(False); ops.set_grad_sinks(None)
    _, g32 = R.transfuse_train_step(R.to_torch(pn), batches_cpu, R.TFState(training=True))
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in R.to_torch(pn).items()}
    _, g64 = R.transfuse_train_step(P64, [(i.double(), l.double(), d) for i, l, d in batches_cpu], R.TFState(training=True))
    big = max(float(v.norm()) for v in g64.values() if v is not None)
    e_hip, e_ref = [], []
    for n, r in g64.items():
        if r is None or float(r.norm()) <= 1e-5 * big or r.numel() == 1:
            continue
        e_hip.append(float((got[n] - r).norm() / r.norm()))
        e_ref.append(float((g32[n].double() - r).norm() / r.norm()))
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    med = (float(np.median(e_hip)), float(np.median(e_ref)))
    p90 = (float(np.quantile(e_hip, 0.9)), float(np.quantile(e_ref, 0.9)))
    print(f"[transfuse gradient error vs fp64 oracle, {gemm_precision} GEMMs] HIP median {med[0]:.2e} p90 {p90[0]:.2e} max {e_hip.max():.2e} | "
          f"fp32 oracle median {med[1]:.2e} p90 {p90[1]:.2e} max {e_ref.max():.2e}  ({len(e_hip)} tensors)")
    # Measured (MI355X, seed 9): fp32 oracle vs fp64 oracle -- median 1.5e-3, p90 5.9e-3, max 7.8e-3: two CORRECT implementations of the same
    # arithmetic already differ by up to 0.8 % per tensor.  HIP with fp32 GEMMs: median 5.6e-3, p90 1.1e-2, max 2.1e-2 (3.7x / 1.9x the oracle's
    # own figures: another summation order in every convolution and BatchNorm statistic); bf16x3 GEMMs (forward moved by 2.7e-4 instead of
    # 1e-6): median 1.8e-2, p90 2.9e-2, max 4.0e-2.  The bounds below are those factors with ~1.6x head room.
    km, kp = (6.0, 3.0) if gemm_precision == "fp32" else (20.0, 8.0)
    assert len(e_hip) > 300
    assert med[0] <= km * med[1] + 1e-4, (med, p90)
    assert p90[0] <= kp * p90[1] + 1e-4, (med, p90)
    assert e_hip.max() <= (3e-2 if gemm_precision == "fp32" else 6e-2), float(e_hip.max())


def test_transfuse_domain_batched_step_equals_per_domain_step():
    """fuse_domains=True (one forward over the concatenated domain batches, BatchNorm statistics per domain batch incl. the
    single-channel BatchNorms, running statistics updated once per domain in order) == one forward per domain"""
    from mdvit_amd.transfuse import transfuse_train_step
    from oracle.gen_golden import synth_image, synth_label
    B, S = 2, 256
    batches = [(synth_image(3200 + d, B, S, S).to(dev()), synth_label(3300 + d, B, S, S).to(dev()), torch.full((B,), d, dtype=torch.long)) for d in (0, 1, 3)]
    res = []
    for fuse in (True, False):
        m, _ = _build(7)
        m.train()
        r = transfuse_train_step(m, batches, fuse_domains=fuse)
        torch.cuda.synchronize()
        res.append((r, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
    check(res[0][0]["per_domain"], res[1][0]["per_domain"], tol=1e-5, name="per-domain losses")
    for k in res[1][2]:
        check(res[0][2][k], res[1][2][k], tol=1e-5, name=k)
    big = max(float(v.double().norm()) for v in res[1][1].values())
    bad = []
    for n, ref in res[1][1].items():
        if float(ref.double().norm()) <= 1e-5 * big:
            continue
        e = float((res[0][1][n].double() - ref.double()).norm() / ref.double().norm())
        if not e <= (0.3 if ref.numel() == 1 else 5e-2):          # same arithmetic, different GEMM tilings at 3x the rows: kink flips only
            bad.append(f"{n}: {e:.2e}")
    assert not bad, bad[:8]


def test_transfuse_bench_step_32_images_fused_equals_per_domain_at_bs8():
    """BASELINE configs[4] as bench.py --model transfuse --batch 8 runs it: the 32-image domain-batched step (4 domains x 8 at 256x256,
    gradient buckets, side stream) against one forward + backward per domain with the same weights: per-domain losses, BatchNorm running
    statistics and every gradient tensor (same arithmetic, other GEMM tilings at 4x the rows: kink flips only -- see the calibration test)"""
    from mdvit_amd import ops
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.transfuse import transfuse_train_step
    from oracle.gen_golden import synth_image, synth_label
    B, S = 8, 256
    batches = [(synth_image(5200 + d, B, S, S).to(dev()), synth_label(5300 + d, B, S, S).to(dev()), torch.full((B,), d, dtype=torch.long)) for d in range(4)]
    res = []
    for fuse in (True, False):
        m, _ = _build(11)
        m.train()
        acc = GradAccumulator(m.parameters()); acc.attach_sinks(); ops.enable_side_stream(True)
        try:
            r = transfuse_train_step(m, batches, accumulator=acc, fuse_domains=fuse)
            torch.cuda.synchronize()
            res.append((r, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                        {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
        finally:
            ops.enable_side_stream(False); ops.set_grad_sinks(None)
    check(res[0][0]["per_domain"], res[1][0]["per_domain"], tol=1e-5, name="per-domain losses")
    for k in res[1][2]:
        check(res[0][2][k], res[1][2][k], tol=1e-5, name=k)
    big = max(float(v.double().norm()) for v in res[1][1].values())
    bad = []
    for n, ref in res[1][1].items():
        assert torch.isfinite(res[0][1][n]).all(), n
        if float(ref.double().norm()) <= 1e-5 * big:
            continue
        e = float((res[0][1][n].double() - ref.double()).norm() / ref.double().norm())
        if not e <= (0.3 if ref.numel() == 1 else 5e-2):
            bad.append(f"{n}: {e:.2e}")
    assert not bad, bad[:8]


def test_transfuse_bench_step_gradients_are_reproducible_run_to_run():
    """the 32-image TransFuse_S_adapt bench step (its CNN branch runs on a stream of its own, weight gradients on the side stream into the
    buckets) three times from one seed: every gradient tensor within 1e-5 relative L2 of the first run (round 4 measured it bit-identical run to run by hand,
    tools/probe/step_determinism.py transfuse)."""
    from mdvit_amd import ops
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.transfuse import transfuse_train_step
    from oracle.gen_golden import synth_image, synth_label
    B, S = 8, 256
    batches = [(synth_image(5200 + d, B, S, S).to(dev()), synth_label(5300 + d, B, S, S).to(dev()), torch.full((B,), d, dtype=torch.long)) for d in range(4)]
    res = []
    for _ in range(3):
        m, _unused = _build(11)
        m.train()
        acc = GradAccumulator(m.parameters()); acc.attach_sinks(); ops.enable_side_stream(True)
        try:
            transfuse_train_step(m, batches, accumulator=acc, fuse_domains=True)
            ops.join_side_stream()
            torch.cuda.synchronize()
            res.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        finally:
            ops.enable_side_stream(False); ops.set_grad_sinks(None)
        del m, acc
        torch.cuda.empty_cache()
    big = max(float(v.double().norm()) for v in res[0].values())
    worst = max((float((g[n].double() - res[0][n].double()).norm()) / float(res[0][n].double().norm()), n)
                for g in res[1:] for n in res[0] if float(res[0][n].double().norm()) > 1e-5 * big)
    assert worst[0] <= 1e-5, f"gradients differ between identical runs: {worst}"


def test_transfuse_thirty_steps_on_one_batch_drive_the_loss_down():
    """end to end with everything the bench uses: domain-batched forward, implicit-GEMM convolutions, MFMA attention, weight gradients on
    the side stream straight into the buckets, the one-launch AdamW -- the summed structure loss of a fixed batch must fall and stay finite"""
    from mdvit_amd import ops
    from mdvit_amd.optim import FusedAdamW
    from mdvit_amd.parallel import GradAccumulator
    from mdvit_amd.transfuse import transfuse_train_step
    from oracle.gen_golden import synth_image, synth_label
    B, S = 2, 256
    batches = [(synth_image(4200 + d, B, S, S).to(dev()), synth_label(4300 + d, B, S, S).to(dev()), torch.full((B,), d, dtype=torch.long)) for d in range(4)]
    m, _ = _build(11, drop=0.1)
    m.train()
    ops.enable_side_stream(True)
    acc = GradAccumulator(m.parameters(), late=[p for n, p in m.named_parameters() if "domain_layer" in n])
    acc.attach_sinks()
    try:
        opt = FusedAdamW(acc, lr=2e-4, weight_decay=0.01)
        losses = []
        for _ in range(30):
            losses.append(transfuse_train_step(m, batches, optimizer=opt, accumulator=acc, fuse_domains=True)["loss"])
        torch.cuda.synchronize()
        losses = [float(l) for l in losses]
    finally:
        ops.enable_side_stream(False)
        ops.set_grad_sinks(None)
    assert all(np.isfinite(l) for l in losses), losses
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:3]), (losses[:3], losses[-5:])
