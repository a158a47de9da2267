"""Data parallelism for the MDViT step: one process per GPU, replicated weights, ONE bucketed
all-reduce of the gradients per optimisation step over RCCL (xGMI), overlapped with the last
backward sweep.  Replaces nn.DataParallel's per-forward parameter broadcast + per-backward
reduce-to-GPU0 (multi_train_MDViT.py:72-74) -- no per-forward traffic at all.

Gradients live as views into a few flat buckets (allocated once), filled in place by autograd.
Buckets are ordered in REVERSE registration order (heads first, stem last) so that they complete
early in the backward sweep; when the last gradient of a bucket has been accumulated during the
final sweep of the step, its all-reduce is issued asynchronously on the process group's stream.
xGMI is point-to-point (7 links/GPU): bucket size defaults to 32 MiB, large enough that RCCL's
reduce-scatter+all-gather stripes every bucket over all links.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params, bucket_bytes: int = 32 << 20, process_group=None, average: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self._armed = False
        self._handles = []
        self._pending: List[int] = []
        # reverse order: parameters used last in forward get their gradients first in backward
        order = list(reversed(self.params))
        self.buckets: List[torch.Tensor] = []
        self._bucket_of = {}
        self._bucket_sizes: List[int] = []
        cur, cur_elems = [], 0
        groups = []
        limit = max(1, bucket_bytes // 4)
        def slot(p):                      # every gradient view starts on a 64-byte boundary (kernels store 16 B vectors)
            return (p.numel() + 15) // 16 * 16

        for p in order:
            if cur and cur_elems + slot(p) > limit:
                groups.append(cur); cur, cur_elems = [], 0
            cur.append(p); cur_elems += slot(p)
        if cur:
            groups.append(cur)
        for bi, grp in enumerate(groups):
            n = sum(slot(p) for p in grp)
            flat = torch.zeros(n, dtype=grp[0].dtype, device=grp[0].device)
            off = 0
            for p in grp:
                p.grad = flat[off:off + p.numel()].view_as(p)
                self._bucket_of[p] = bi
                off += slot(p)
            self.buckets.append(flat)
            self._bucket_sizes.append(len(grp))
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # ---- step protocol --------------------------------------------------------------------------
    def zero_grad(self):
        for b in self.buckets:
            b.zero_()

    def arm(self):
        """Call right before the LAST backward() of the step: from now on a bucket is reduced as soon
        as each of its gradients has been accumulated once more."""
        self._armed = True
        self._pending = list(self._bucket_sizes)
        self._handles = []

    def _on_grad(self, p):
        if not self._armed:
            return
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        if self.world > 1:
            self._handles.append(dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the in-flight reductions; reduce buckets whose parameters got no gradient in the
        last sweep (e.g. unused heads); apply the 1/world average."""
        if self._armed:
            for bi, left in enumerate(self._pending):
                if left > 0:
                    self._launch(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        self._armed = False
        if self.world > 1 and self.average:
            for b in self.buckets:
                b.div_(self.world)

    def check_views(self):
        """autograd must keep accumulating into the bucket views (it does as long as .grad is never re-bound)."""
        for p in self.params:
            bi = self._bucket_of[p]
            b = self.buckets[bi]
            if p.grad is None or p.grad.data_ptr() < b.data_ptr() or p.grad.data_ptr() >= b.data_ptr() + b.numel() * b.element_size():
                return False
        return True


class GradAccumulator:
    """Gradient accumulation over the 4 domains x 2 sweeps of one step with a handful of kernels, and the data-parallel
    all-reduce of the buckets UNDERNEATH the rest of the backward.

    autograd's AccumulateGrad issues one small add per parameter per backward() (432 x 7 per step).  Here every
    sweep runs with p.grad = None and its fresh gradients are folded into flat buckets by ONE multi-tensor add (most
    weight gradients never reach autograd: the wgrad kernels add them into the buckets directly, see attach_sinks).

    Overlap (world_size > 1).  `late`: parameters that still receive gradients after the heavy sweep -- in the merged
    two-sweep step (train.mdvit_train_step, full sweep first) only the domain adapters do (the data-gradient-only aux
    sweep hands them minus their aux gradient).  They get buckets of their own at the END of the bucket list, so every
    other bucket is final as soon as the full sweep's kernels are done: end_sweep(last=False, remaining=late) then issues
    the all-reduce of those buckets on the process group's stream, ordered after the side stream's weight-gradient
    kernels and the main stream's fold -- it runs over xGMI while the main stream works through the aux sweep.  The late
    buckets (a few hundred KB) follow after the last sweep; finish() makes the main stream wait and applies 1/world.
    Sweep orders in which the last sweep still touches everything (the reference's literal two sweeps) reduce all
    buckets at the end -- same code path, no overlap.
    (GradBucketReducer above is the hook-driven variant for a plain autograd backward.)"""

    def __init__(self, params, bucket_bytes: int = 32 << 20, process_group=None, average: bool = True, late=None):
        params = [p for p in params if p.requires_grad]
        late_ids = {id(p) for p in (late or [])}
        # GradBucketReducer buckets in REVERSE registration order: put the late parameters first so they end up in the last buckets,
        # and cut a bucket boundary between the two classes
        early = [p for p in params if id(p) not in late_ids]
        latep = [p for p in params if id(p) in late_ids]
        if not early:                          # every parameter is "late": nothing can go on the wire early -- one class, no split
            early, latep = latep, []
        self.reducer = GradBucketReducer(early, bucket_bytes, process_group, average)
        self._n_early = len(self.reducer.buckets)
        if latep:
            lr = GradBucketReducer(latep, bucket_bytes, process_group, average)
            for h in lr._hooks:
                h.remove()
            off = len(self.reducer.buckets)
            self.reducer.buckets += lr.buckets
            self.reducer._bucket_sizes += lr._bucket_sizes
            for p, bi in lr._bucket_of.items():
                self.reducer._bucket_of[p] = bi + off
            self.reducer.params += lr.params
        for h in self.reducer._hooks:          # this class drives the reduction itself; the per-parameter hooks stay off
            h.remove()
        self.reducer._hooks = []
        self.params = self.reducer.params
        self.views = [p.grad for p in self.params]          # views into the flat buckets
        for p in self.params:
            p.grad = None
        self.sinks = {p: v for p, v in zip(self.params, self.views) if p.is_cuda}
        self._handles = []
        self._reduced = [False] * len(self.reducer.buckets)
        self.overlapped_buckets = 0          # how many buckets of the last step were reduced before its last sweep ended (tests / bench)

    def attach_sinks(self, flag: bool = True):
        """Let the HIP wgrad kernels add weight gradients straight into the buckets (mdvit_amd.ops.set_grad_sinks)."""
        from . import ops
        ops.set_grad_sinks(self.sinks if flag else None)

    def view_of(self, p):
        """the bucket view that holds p's gradient"""
        for q, v in zip(self.params, self.views):
            if q is p:
                return v
        raise KeyError("not a parameter of this accumulator")

    @property
    def world(self):
        return self.reducer.world

    def _collective(self) -> bool:
        from . import ops as _ops
        return self.world > 1 or (_ops._force_collectives and dist.is_available() and dist.is_initialized())

    def zero(self):
        self.reducer.zero_grad()
        for p in self.params:
            p.grad = None
        self._handles = []
        self._reduced = [False] * len(self.reducer.buckets)
        self.overlapped_buckets = 0

    def begin_sweep(self, last: bool):
        # Every sweep runs with p.grad = None and is folded into the buckets by one multi-tensor add; weight gradients that
        # the HIP kernels accumulate straight into the buckets (sinks, possibly on a side stream) never pass through autograd.
        for p in self.params:
            p.grad = None

    def _fold(self):
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is not None:
                dst.append(v); src.append(p.grad)
        if dst:
            torch._foreach_add_(dst, src)

    def _launch(self, indices):
        """all-reduce of the given buckets, issued from a stream that is ordered after BOTH the main stream (folds) and the side
        stream (weight-gradient kernels writing the sinks); async: the main stream goes on"""
        from . import ops as _ops
        side = _ops._side_stream
        if not self.reducer.buckets:
            return
        main = torch.cuda.current_stream() if torch.cuda.is_available() and self.reducer.buckets[0].is_cuda else None
        ctx = None
        if side is not None and main is not None:
            side.wait_stream(main)             # the fold above (and everything before it) precedes the collective
            ctx = torch.cuda.stream(side)      # ProcessGroupNCCL orders its stream after the CURRENT stream: the side stream's tail
            ctx.__enter__()
        try:
            for bi in indices:
                if not self._reduced[bi]:
                    self._handles.append(dist.all_reduce(self.reducer.buckets[bi], op=dist.ReduceOp.SUM, group=self.reducer.group, async_op=True))
                    self._reduced[bi] = True
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def end_sweep(self, last: bool, remaining=None):
        """remaining: the parameters that later sweeps of this step can still touch (None: unknown -> every bucket stays open
        until the last sweep).  Buckets holding none of them are final now and go on the wire."""
        self._fold()
        if self._collective():
            if last:
                self._launch(range(len(self.reducer.buckets)))
            elif remaining is not None:
                open_b = {self.reducer._bucket_of[p] for p in remaining if p in self.reducer._bucket_of}
                early = [bi for bi in range(len(self.reducer.buckets)) if bi not in open_b]
                self._launch(early)
                self.overlapped_buckets = len(early)
        if last:
            self.finish()
            for p, v in zip(self.params, self.views):       # hand the accumulated gradients to the optimizer
                p.grad = v
        else:
            for p in self.params:
                p.grad = None

    def finish(self):
        """the main stream waits for the collectives; 1/world average"""
        if not self._handles:
            return
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.reducer.average and self.world > 1:
            torch._foreach_div_(self.reducer.buckets, float(self.world))


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None):
    """One-off weight/buffer sync at start-up (replicas are seeded identically; this makes it certain)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
