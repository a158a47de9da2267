"""The C-level block entry (csrc/block.hip: one host call per SerialBlock_adapt pass, mdvit.py:346-361) against the operator-level path it
replaces -- the same kernels in the same order, so outputs and gradients must agree bit for bit (window-weight gradients, which use
LDS float atomics, to 1e-6)."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def make_stage(C, ratio, adapt, drop=0.1, droppath=0.1, seed=0):
    from mdvit_amd.blocks import MHSA_stage_adapt, init_weights_
    torch.manual_seed(seed)
    st = MHSA_stage_adapt(256, C, num_layers=2, num_heads=8, mlp_ratio=ratio, qkv_bias=True, drop_rate=drop, drop_path_rate=droppath,
                          adapt_method="Sup" if adapt else None)
    init_weights_(st)
    with torch.no_grad():
        for p in st.parameters():                      # biases and norms away from their 0 / 1 initial values
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    return st.to(dev()).train()


def run(st, x, label, H, W, g, entry, monkeypatch, dgrad_only=False):
    from mdvit_amd import ops
    monkeypatch.setattr(ops, "_block_entry", entry)
    monkeypatch.setattr(ops, "_key_counter", itertools.count(41))
    torch.manual_seed(7)                               # the DropPath draws
    for p in st.parameters():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    y = st(xin, H, W, label)
    ops.set_dgrad_only(dgrad_only)
    try:
        y.backward(g)
    finally:
        ops.set_dgrad_only(False)
    ops.join_side_stream()
    torch.cuda.synchronize()
    return y.detach(), xin.grad, {n: (None if p.grad is None else p.grad.clone()) for n, p in st.named_parameters()}


@pytest.mark.parametrize("C,ratio,adapt", [(64, 8, True), (128, 8, True), (320, 4, True), (512, 4, False), (64, 8, False)])
def test_block_entry_equals_operator_path(C, ratio, adapt, gemm_precision, monkeypatch):
    from mdvit_amd import ops
    B, H, W = 3, 12, 20
    x = torch.randn(B, H * W, C, device=dev())
    label = torch.nn.functional.one_hot(torch.tensor([1, 3, 0]), 4).float().to(dev()) if adapt else None
    g = torch.randn(B, H * W, C, device=dev())
    st = make_stage(C, ratio, adapt)
    ref = run(st, x, label, H, W, g, False, monkeypatch)
    got = run(st, x, label, H, W, g, True, monkeypatch)
    assert torch.equal(got[0], ref[0]), f"y differs by {float((got[0] - ref[0]).abs().max()):.3e}"
    assert torch.equal(got[1], ref[1]), f"dx differs by {float((got[1] - ref[1]).abs().max()):.3e}"
    def close(a, b):
        return float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-12)

    for n in ref[2]:
        a, b = got[2][n], ref[2][n]
        assert (a is None) == (b is None), n
        if a is None:
            continue
        # float atomics (LDS adds inside the window / depthwise weight-gradient tiles and the adapter's e; the fp32 mode's bias column sums):
        # last-bit differences run to run.  Everything else is bitwise reproducible, so it must be bitwise EQUAL
        if "crpe" in n or "cpe" in n or "domain_layer" in n or (gemm_precision == "fp32" and n.endswith(".bias")):
            assert close(a, b), n
        else:
            assert torch.equal(a, b), f"{n} differs by {float((a - b).abs().max()):.3e}"
    # the data-gradient-only sweep: dx, and the adapters' NEGATED gradients alone
    ref = run(st, x, label, H, W, g, False, monkeypatch, dgrad_only=True)
    got = run(st, x, label, H, W, g, True, monkeypatch, dgrad_only=True)
    assert torch.equal(got[1], ref[1])
    for n in ref[2]:
        a, b = got[2][n], ref[2][n]
        assert (a is None) == (b is None), n
        if a is not None:
            assert "domain_layer" in n and close(a, b), n


def test_block_entry_with_gradient_buckets_and_side_stream(monkeypatch):
    """weight gradients accumulated straight into persistent buffers by the side stream's kernels (parallel.GradAccumulator sinks)"""
    from mdvit_amd import ops
    B, H, W, C = 4, 16, 16, 64
    x = torch.randn(B, H * W, C, device=dev())
    label = torch.nn.functional.one_hot(torch.tensor([0, 1, 2, 3]), 4).float().to(dev())
    g = torch.randn(B, H * W, C, device=dev())
    st = make_stage(C, 8, True)
    # (the C = 64 MLP backward runs as ONE kernel where no weight-gradient stream exists and as data-gradient + weight-gradient kernels where one does: dx then differs in
    #  the summation order over the hidden axis.  This test compares the two paths bit for bit, so both take the two-kernel form; the one-kernel form has its own test.)
    monkeypatch.setattr(ops, "_mlp_rc_bwd", "0")
    from mdvit_amd._lib import call
    call("mdvit_block_config", 0)
    ref = run(st, x, label, H, W, g, False, monkeypatch)
    sinks = {p: torch.full_like(p, 0.25) for p in st.parameters()}
    ops.enable_side_stream(True)
    ops.set_grad_sinks(sinks)
    try:
        for _ in range(3):                             # repeated: a missing cross-stream dependency shows as run-to-run drift
            for v in sinks.values():
                v.fill_(0.25)
            got = run(st, x, label, H, W, g, True, monkeypatch)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
            for n, p in st.named_parameters():
                total = sinks[p] - 0.25 + (p.grad if p.grad is not None else 0)
                tol = 2e-6 if ("crpe" in n or "cpe" in n) else 1e-6
                assert float((total - ref[2][n]).abs().max()) <= tol * max(float(ref[2][n].abs().max()), 1e-12), n
    finally:
        ops.set_grad_sinks(None)
        ops.enable_side_stream(False)
        call("mdvit_block_config", 1)


def test_block_entry_first_adapter_of_the_aux_sweep(monkeypatch):
    from mdvit_amd import ops
    B, H, W, C = 2, 8, 8, 64
    x = torch.randn(B, H * W, C, device=dev())
    label = torch.nn.functional.one_hot(torch.tensor([2, 1]), 4).float().to(dev())
    g = torch.randn(B, H * W, C, device=dev())
    st = make_stage(C, 8, True, drop=0.0, droppath=0.0)
    st.mhca_blks[0].factoratt_crpe.aux_first = True
    ref = run(st, x, label, H, W, g, False, monkeypatch, dgrad_only=True)
    got = run(st, x, label, H, W, g, True, monkeypatch, dgrad_only=True)
    # the sweep ends at the first adapter: the block entry hands nothing on (the operator path still carries the residual branch's gradient
    # to the block input, where the model's ops.aux_stop drops it -- model._trunk)
    assert got[1] is None
    for n in ref[2]:
        a, b = got[2][n], ref[2][n]
        assert (a is None) == (b is None), n
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-12), n


@pytest.mark.parametrize("entry", [False, True])
@pytest.mark.parametrize("dgrad_only", [False, True])
def test_adapters_owned_by_one_node_give_the_blocks_own_adapter_gradients(entry, dgrad_only, monkeypatch):
    """Inside ops.da_precomputed the blocks take their adapter output from the all-adapters node (_DaMany) and hand e = a * dL/da back to it (MdvitBlockGrads.e_out /
    _FactorAtt's a_pre input) instead of running mdvit_da_fwd / mdvit_da_bwd themselves: same output and data gradient bit for bit, every parameter gradient the
    same (e is summed with float adds in LDS: last-bit differences run to run in either mode), in the full and in the data-gradient-only sweep (first adapter included)."""
    from mdvit_amd import ops
    B, H, W, C = 3, 12, 20, 64
    x = torch.randn(B, H * W, C, device=dev())
    label = torch.nn.functional.one_hot(torch.tensor([2, 1, 3]), 4).float().to(dev())
    g = torch.randn(B, H * W, C, device=dev())
    st = make_stage(C, 8, True)
    st.mhca_blks[0].factoratt_crpe.aux_first = True
    ref = run(st, x, label, H, W, g, entry, monkeypatch, dgrad_only=dgrad_only)
    ads = []
    for blk in st.mhca_blks:
        att = blk.factoratt_crpe
        d0, d2 = att.domain_layer[0], att.domain_layer[2]
        ads.append((d0.weight, d0.bias, d2.weight, d2.bias, att.num_heads))
    launches = []
    orig = ops._DaMany.backward
    monkeypatch.setattr(ops._DaMany, "backward", staticmethod(lambda ctx, *es: (launches.append(sum(e is not None for e in es)), orig(ctx, *es))[1]))
    with ops.da_precomputed(label, ads):
        got = run(st, x, label, H, W, g, entry, monkeypatch, dgrad_only=dgrad_only)
    assert launches == [len(ads)]                       # one backward for every adapter of the stage
    assert torch.equal(got[0], ref[0])
    if ref[1] is None or got[1] is None:
        assert got[1] is None or dgrad_only          # (the operator path hands the residual branch's gradient on in the aux sweep; model._trunk drops it)
    else:
        assert torch.equal(got[1], ref[1])
    for n in ref[2]:
        a, b = got[2][n], ref[2][n]
        assert (a is None) == (b is None), n
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-12), n


def test_mixed_mode_stores_the_c128_mlp_hidden_tensors_as_bf16(monkeypatch):
    """MdvitBlockDesc.store_bf16 (the bf16 / "mixed" mode, BASELINE configs[3]): h = drop1(gelu(u)) of the forward and du of the backward -- the two
    [tokens, hidden] tensors the C = 128 block still moves, operands of the fc2 / fc1 weight-gradient GEMMs only -- live in HBM as bf16.  What is stored is
    the hi plane the kernels form for their own second product: y, dx and every other gradient are IDENTICAL to fp32 storage; fc1.weight, fc1.bias and
    fc2.weight see bf16-rounded operands (2^-9 relative per element, averaging out over the token sum).  The saved buffer shrinks by 4 of its 17 C floats
    per token."""
    from mdvit_amd import ops
    B, H, W, C = 3, 12, 20, 128
    x = torch.randn(B, H * W, C, device=dev())
    label = torch.nn.functional.one_hot(torch.tensor([1, 3, 0]), 4).float().to(dev())
    g = torch.randn(B, H * W, C, device=dev())
    st = make_stage(C, 8, True)
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    try:
        monkeypatch.setattr(ops, "_store_bf16", False)
        ref = run(st, x, label, H, W, g, True, monkeypatch)
        monkeypatch.setattr(ops, "_store_bf16", True)
        got = run(st, x, label, H, W, g, True, monkeypatch)
        got2 = run(st, x, label, H, W, g, True, monkeypatch)
        # the saved buffer really is smaller: hidden / 2 floats per token less
        sizes = []
        for flag in (False, True):
            monkeypatch.setattr(ops, "_store_bf16", flag)
            y = st(x.clone().requires_grad_(True), H, W, label)
            sizes.append(y.grad_fn.saved_tensors[1].numel())
            del y
        assert sizes[0] - sizes[1] >= B * H * W * 1024 // 2 - 256, sizes
    finally:
        ops.set_gemm_precision(prev)
    assert torch.equal(got[0], ref[0]), f"y differs by {float((got[0] - ref[0]).abs().max()):.3e}"
    assert torch.equal(got[1], ref[1]), f"dx differs by {float((got[1] - ref[1]).abs().max()):.3e}"
    moved = []
    for n in ref[2]:
        a, b = got[2][n], ref[2][n]
        assert (a is None) == (b is None), n
        if a is None:
            continue
        rel = float((a.double() - b.double()).norm()) / max(float(b.double().norm()), 1e-30)
        if n.endswith("mlp.fc1.weight") or n.endswith("mlp.fc1.bias") or n.endswith("mlp.fc2.weight"):
            assert 0 < rel <= 4e-3, f"{n}: bf16-stored operand moved the gradient by {rel:.2e} (expected ~1e-3)"
            moved.append(n)
            assert torch.equal(a, got2[2][n]), f"{n}: not repeatable"
        elif "crpe" in n or "cpe" in n or "domain_layer" in n:
            assert rel <= 2e-6, n
        else:
            assert torch.equal(a, b), f"{n} differs by {rel:.2e}"
    assert len(moved) == 6, moved          # two blocks x (fc1.weight, fc1.bias, fc2.weight)
