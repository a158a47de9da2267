"""The optimisation step of multi_train_MDViT.py:129-213 (and multi_train_BASE.py:150-200) for the
HIP-backed models: per domain forward -> fused losses -> the "det_Sup" two-sweep backward
(aux loss with every `domain_layer` parameter frozen, then alpha*kt + (1-alpha)*loss into everything),
gradients accumulated over the domains, optional data-parallel all-reduce, optimizer step.

Because the step loss is a sum over domains, back-propagating each domain right after its forward
gives the same gradients as the reference's "4 forwards, then 2 sweeps" while keeping only one
domain's activations alive; `per_domain_backward=False` reproduces the reference's
order exactly.
"""
from __future__ import annotations

import functools
import operator
import os

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops
from .losses import domain_losses, seg_loss
from .parallel import GradAccumulator, GradBucketReducer


_GROUPED_LOSSES = os.environ.get("MDVIT_GROUPED_LOSSES", "1") != "0"      # the G domain batches' losses in one launch each way (0: one op per domain + additions, A/B)
_AUX_STREAM_FORCE = os.environ.get("MDVIT_AUX_SWEEP_STREAM_FORCE", "0") == "1"      # probes only: the aux sweep on its own stream whatever the model declares
_two_stream_sweeps = os.environ.get("MDVIT_SWEEP_STREAMS", "1") != "0"      # A/B switch: 0 = both sweeps on the main stream
_EXP_SKIP_AUX_SWEEP = os.environ.get("MDVIT_EXP_SKIP_AUX_SWEEP", "0") == "1"      # an EXPERIMENT (wrong adapter gradients): what the data-gradient-only sweep costs the step
_timeline = None      # tools/sweep_timeline.py: a list here collects (tag, event, host seconds) at the sweeps' stream ends


def _tl(tag, stream=None):
    if _timeline is not None:
        import time
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        _timeline.append((tag, e, time.perf_counter()))


def _backward(loss, **kw):
    ops.backward(loss, **kw)


def _da_params(model):
    """the domain adapters' parameters (multi_train_MDViT.py:198-200 walks named_parameters() twice per step; here the walk -- ~1 ms of host time for the 432 tensors --
    runs once per model object: Module.to / .cuda / load_state_dict keep the Parameter objects, so the list stays valid; `del model._mdvit_da_params` after surgery)"""
    hit = model.__dict__.get("_mdvit_da_params")
    if hit is not None:
        return hit
    da = [p for n, p in model.named_parameters() if "domain_layer" in n]
    model.__dict__["_mdvit_da_params"] = da          # (a plain attribute: not a registered submodule / buffer, never in the state_dict)
    return da


def mdvit_train_step(model, batches: Sequence[tuple], optimizer=None, alpha: float = 0.5, num_domains: int = 4,
                     reducer: Optional[GradBucketReducer] = None, per_domain_backward: bool = True,
                     use_domain_label: bool = True, accumulator: Optional[GradAccumulator] = None,
                     merged_sweeps: bool = False, fuse_domains: int = 1, phase_events: Optional[list] = None,
                     with_metrics: bool = False) -> Dict[str, torch.Tensor]:
    """batches: [(img (B,3,H,W), label (B,1,H,W), set_id (B,) int64)] one per domain.
    Returns the summed losses as device tensors (no host sync inside the step).
    accumulator: fused gradient accumulation (+ overlapped all-reduce when world_size > 1), see parallel.GradAccumulator.
    merged_sweeps: same gradients as the reference's two sweeps with half the weight-gradient work.  Back-propagation
      is linear in the upstream gradient, so  grad(aux, adapters frozen) + grad(uni)  ==  grad(aux + uni) - [adapter part of
      grad(aux)].  Sweep 1 therefore runs data-gradients-only (no wgrad GEMMs / reductions) and hands every domain adapter
      MINUS its aux gradient; sweep 2 is one ordinary backward of aux + uni."""
    da = _da_params(model)
    fused_forward = fuse_domains > 1 and len(batches) > 0 and fuse_domains >= len(batches)       # one forward per step: every module is used once
    ops.refresh_transposes()          # the W^T copies of the bf16x3 data-gradient GEMMs follow the last optimizer update (one launch)
    if accumulator is not None:
        accumulator.zero()
    elif reducer is not None:
        reducer.zero_grad()
    elif optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    else:
        model.zero_grad(set_to_none=True)
    if fuse_domains > 1:
        batches = _fuse_batches(batches, fuse_domains, num_domains, use_domain_label)

    def sweep(loss, last, retain=False, join=True, remaining=None, on_stream=None):
        if accumulator is not None:
            accumulator.begin_sweep(last)
        elif last and reducer is not None:
            reducer.arm()
        if on_stream is None:
            _backward(loss, retain_graph=retain)
        else:
            # the whole sweep on a stream of its own (ops.set_sweep_stream): every node of it is one of our Functions (ops.fork at the
            # trunk's multi-consumer tensors) and the root gradient is handed in, so nothing of it is launched on the main stream
            main = ops.current_stream_obj()
            _tl("aux sweep: runnable (its stream)", on_stream)
            ops.set_sweep_stream(on_stream)
            try:
                _backward(loss, retain_graph=retain, gradient=ops.one_like(loss))
            finally:
                ops.set_sweep_stream(None)
            _tl("aux sweep: last kernel (its stream)", on_stream)
            ops.stream_wait(main, on_stream)
        if on_stream is None:
            _tl("sweep on main: last kernel")
        if join:
            if ops._side_stream is not None:
                _tl("side stream: last weight-gradient kernel", ops._side_stream)
            ops.join_side_stream()      # weight gradients may have been produced on the side stream
        if accumulator is not None:
            accumulator.end_sweep(last, remaining=remaining)

    def two_sweeps(aux_sum, uni, last):
        if merged_sweeps:
            if accumulator is not None and os.environ.get("MDVIT_SWEEP_ORDER", "full_first") == "full_first":
                # the ordinary backward of aux + uni FIRST, and no join after it: its weight-gradient kernels (side stream, straight
                # into the bucket sinks) then drain underneath the whole data-gradient-only sweep instead of piling up behind the
                # last kernels of the step.  (The two sweeps commute; what this sweep hands autograd is folded into the buckets
                # on the main stream and touches other bucket elements than the sinks.)
                # Data parallel: what the aux sweep can still touch is the domain adapters only -- on the last domain forward of the
                # step every other gradient bucket is final after this sweep and its all-reduce is issued underneath the aux sweep.
                # Two streams: the data-gradient-only aux sweep does not depend on the full sweep, only on the forward -- it runs on a
                # stream of its own, ordered after the forward (the event below) and joined before the gradients are folded.
                # (only when the aux graph holds nothing the autograd ENGINE would launch itself -- see _aux_graph_is_ours)
                s2 = ops.sweep_stream() if (_two_stream_sweeps and fused_forward and aux_sum.is_cuda and (_AUX_STREAM_FORCE or _aux_graph_is_ours(model, aux_sum))) else None
                if s2 is not None:
                    ops.stream_wait(s2, ops.current_stream_obj())          # recorded BEFORE the full sweep is enqueued
                ops.take_cache_fill_flag()
                sweep(aux_sum + uni, False, retain=True, join=False, remaining=(da if last else None))
                if s2 is not None and ops.take_cache_fill_flag():
                    # a cold step: the full sweep just (re)built derived-weight cache entries (W^T / planes / conv layouts) on the main
                    # stream; the aux sweep would hit them by host-side tag with no stream ordering -- order it after the fills
                    ops.stream_wait(s2, ops.current_stream_obj())
                if _EXP_SKIP_AUX_SWEEP:
                    if accumulator is not None:          # (the end-of-step bookkeeping only)
                        accumulator.begin_sweep(last)
                    ops.join_side_stream()
                    if accumulator is not None:
                        accumulator.end_sweep(last)
                    return
                ops.set_dgrad_only(True)
                try:
                    sweep(aux_sum, last, on_stream=s2)
                finally:
                    ops.set_dgrad_only(False)
                return
            ops.set_dgrad_only(True)
            try:
                sweep(aux_sum, False, retain=True)
            finally:
                ops.set_dgrad_only(False)
            sweep(aux_sum + uni, last)
            return
        for p in da:
            p.requires_grad = False
        sweep(aux_sum, False, retain=True)
        for p in da:
            p.requires_grad = True
        sweep(uni, last)

    def mark(tag):          # optional phase timing (bench.py): an event per phase boundary on the main stream
        if phase_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            phase_events.append((tag, e))

    tot = tot_aux = tot_kt = None
    stash = []
    metric_rows = []        # per domain [dice, iou, aux dice, aux iou] of the thresholded outputs, kept on the device
    mark("start")
    for i, batch in enumerate(batches):
        img, label, set_id = batch[0], batch[1], batch[2]
        # set_id is a HOST tensor, as from the DataLoader (multi_train_MDViT.py:137-139): reading set_id[0] must not
        # synchronise the GPU stream -- the host keeps enqueueing the next domain while the GPU works on this one
        if set_id.is_cuda:
            set_id = set_id.cpu()
        G = batch[4] if len(batch) > 4 else 1         # domain batches fused into this forward (see _fuse_batches)
        Bd = img.shape[0] // G
        d = str(int(set_id[0])) if G == 1 else [str(int(set_id[g * Bd])) for g in range(G)]
        if use_domain_label:
            # an optional 4th entry carries the one-hot label already on the device (needed under HIP-graph capture)
            domain_label = batch[3] if len(batch) > 3 and batch[3] is not None else \
                F.one_hot(set_id, num_domains).float().to(img.device, non_blocking=True)
            out, aux = model(img, domain_label, d)
        else:
            out, aux = model(img, d=d)
        if with_metrics:       # multi_train_MDViT.py:172-179, without the per-domain .cpu().numpy()
            for g in range(G):
                metric_rows.append(ops.seg_metrics(out[g * Bd:(g + 1) * Bd], None if aux is None else aux[g * Bd:(g + 1) * Bd],
                                                   label[g * Bd:(g + 1) * Bd])[0])
        if G == 1:
            l, la, lk = domain_losses(out, aux, label)
        else:       # per-domain BCE/Dice/KT (each a mean over ITS batch, multi_train_MDViT.py:147-153), then summed
            if _GROUPED_LOSSES and aux is not None:
                l, la, lk = ops.seg_losses_groups(out, aux, label, G)
            else:
                og, ag = ops.split_groups(out, G), ops.split_groups(aux, G)
                per = [domain_losses(og[g], ag[g], label[g * Bd:(g + 1) * Bd]) for g in range(G)]
                l, la, lk = (functools.reduce(operator.add, [t[j] for t in per]) for j in range(3))          # (sum() starts from the int 0: one more add launch per loss)
        tot = l.detach() if tot is None else tot + l.detach()
        tot_aux = la.detach() if tot_aux is None else tot_aux + la.detach()
        tot_kt = lk.detach() if tot_kt is None else tot_kt + lk.detach()
        mark("fwd")
        if per_domain_backward:
            two_sweeps(la, alpha * lk + (1 - alpha) * l, last=(i == len(batches) - 1))
            mark("bwd")
        else:
            stash.append((l, la, lk))
    if not per_domain_backward:
        tot3 = [functools.reduce(operator.add, [s[j] for s in stash]) for j in range(3)]
        two_sweeps(tot3[1], alpha * tot3[2] + (1 - alpha) * tot3[0], last=True)
        mark("bwd")
    if accumulator is None and reducer is not None:
        reducer.finish()
    if optimizer is not None:
        optimizer.step()
        mark("opt")
    res = {"loss": tot, "aux_loss": tot_aux, "kt_loss": tot_kt}
    if with_metrics:
        res["metrics"] = torch.stack(metric_rows)          # [domains, 4]
    return res


def _aux_graph_is_ours(model, aux_sum) -> bool:
    """May the data-gradient-only aux sweep run on a stream of its own?  Only if every kernel of that backward is launched by this package's Functions (which
    follow ops.set_sweep_stream): autograd's own gradient accumulation at a tensor with several consumers, or a torch op's backward, is launched by the engine on
    the stream the FORWARD ran on -- unordered with a sweep that was moved elsewhere.  (Found with the DeepLabV3 peer heads before ASPP's five-consumer input went
    behind ops.fork: the adapters' gradients ~100 % off in one of eight cold steps.)  The graph's structure does not change from step to step: audited once per
    model object (ops.audit_sweep_graph, a Python walk of ~1000 nodes), the verdict and the findings cached on it."""
    # the verdict is cached per (model object, what shapes its aux graph): the peer-head family, train / eval, how many domain batches the forward fused, whether
    # gradient sinks are attached -- a swapped head, another fusion width or a detached accumulator is audited again (the cache used to live as long as the object)
    # (aux_sum is a scalar -- its shape says nothing; the fusion width is the number of loss groups / peer heads its node carries, and WHICH parameters have
    #  sinks is the identity of the sink table, not whether one exists)
    sig = (getattr(model, "decoder_name", None), bool(model.training), type(aux_sum.grad_fn).__name__, len(getattr(aux_sum.grad_fn, "next_functions", ())),
           _fusion_width(aux_sum), id(ops._sinks), len(ops._sinks),
           tuple(type(getattr(model, f"debranch{i}", None)).__name__ for i in range(1, 5)))
    cache = getattr(model, "_aux_sweep_graph_cache", None)
    if cache is not None and cache[0] == sig:
        return cache[1]
    native, fanin = ops.audit_sweep_graph(aux_sum)
    ok = not native and not fanin
    try:
        model._aux_sweep_graph_cache = (sig, ok)
        model._aux_sweep_graph_ok = ok
        model._aux_sweep_graph_findings = (native, fanin)
    except Exception:           # (an object that refuses attributes: audit every step)
        pass
    return ok


def _fusion_width(aux_sum) -> int:
    """how many domain batches the forward behind this aux loss fused: the group count of the grouped loss node (ops._SegLossesGroups) when there is one, else 1"""
    seen, stack = set(), [aux_sum.grad_fn]
    for _ in range(64):          # the loss node sits within a few additions / multiplications of the root
        if not stack:
            break
        fn = stack.pop()
        if fn is None or id(fn) in seen:
            continue
        seen.add(id(fn))
        G = getattr(fn, "G", None)
        if isinstance(G, int):
            return G
        stack.extend(nf for nf, _ in getattr(fn, "next_functions", ()))
    return 1


def _fuse_batches(batches, fuse_domains, num_domains, use_domain_label):
    """Concatenate runs of up to `fuse_domains` equally shaped domain batches into one domain-batched forward
    (model(img, label, [d0, d1, ...])): same losses and gradients as separate forwards (BatchNorm statistics are kept
    per domain batch), a quarter of the kernel launches and better-filled kernels at small per-domain batch sizes.
    A batch that is already fused (5-tuple) passes through."""
    out, run = [], []

    def flush():
        if not run:
            return
        if len(run) == 1:
            out.append(run[0])
        else:
            img = torch.cat([b[0] for b in run], 0)
            lab = torch.cat([b[1] for b in run], 0)
            sid = torch.cat([b[2].cpu() for b in run], 0)
            dl = None
            if use_domain_label:
                dl = torch.cat([b[3] for b in run], 0) if all(len(b) > 3 and b[3] is not None for b in run) else \
                    F.one_hot(sid, num_domains).float().to(img.device, non_blocking=True)
            out.append((img, lab, sid, dl, len(run)))
        run.clear()

    for b in batches:
        if len(b) > 4 or (run and (b[0].shape != run[0][0].shape or len(run) >= fuse_domains)):
            flush()
        if len(b) > 4:
            out.append(b)
        else:
            run.append(b)
    flush()
    return out


def base_train_step(model, batches: Sequence[tuple], optimizer=None, reducer: Optional[GradBucketReducer] = None,
                    num_domains: int = 4, use_domain_label: bool = False,
                    accumulator: Optional[GradAccumulator] = None) -> Dict[str, torch.Tensor]:
    """multi_train_BASE.py:150-200: per domain loss = BCE + Dice, one backward of the sum."""
    ops.refresh_transposes()          # cached W^T / weight planes follow the last optimizer update (one launch each)
    if accumulator is not None:
        accumulator.zero()
    elif reducer is not None:
        reducer.zero_grad()
    elif optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    else:
        model.zero_grad(set_to_none=True)
    tot = None
    for i, (img, label, set_id, *pre) in enumerate(batches):          # pre[0]: the one-hot domain label already on the device (graph capture)
        if use_domain_label:
            out = model(img, pre[0] if pre and pre[0] is not None else F.one_hot(set_id.cpu(), num_domains).float().to(img.device, non_blocking=True))
        else:
            out = model(img)
        l = seg_loss(out, label)
        last = i == len(batches) - 1
        if accumulator is not None:
            accumulator.begin_sweep(last)
        elif reducer is not None and last:
            reducer.arm()
        _backward(l)
        ops.join_side_stream()
        if accumulator is not None:
            accumulator.end_sweep(last)
        tot = l.detach() if tot is None else tot + l.detach()
    if accumulator is None and reducer is not None:
        reducer.finish()
    if optimizer is not None:
        optimizer.step()
    return {"loss": tot}
