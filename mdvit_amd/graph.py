"""Whole-step HIP-graph capture: the 4 forwards, the fused losses, both backward sweeps, the gradient
accumulation and the AdamW update of one optimisation step (~6000 kernel launches) are captured once and
replayed, which removes the host launch floor (~110 ms/step of Python + launch overhead at any batch size).

Requirements met elsewhere: no host synchronisation inside the step (set_id lives on the host), dropout masks are
re-keyed on the device (ops.enable_device_seed / ops.bump_seed), DropPath draws come from torch's graph-safe
generator, weight gradients accumulate into static buckets (parallel.GradAccumulator sinks), the optimizer is
`capturable`.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.nn.functional as F

from . import ops


class GraphedStep:
    def __init__(self, step_fn: Callable[[Sequence[tuple]], dict], example_batches: Sequence[tuple], num_domains: int = 4,
                 warmup: int = 2, fuse_domains: int = 1):
        """step_fn(batches) -> dict of device tensors; example_batches fixes shapes and the domain order.
        fuse_domains: the static input buffers are laid out as the domain-batched forwards train._fuse_batches builds,
        so the captured step contains no concatenation; incoming per-domain batches are copied into their slices."""
        from .train import _fuse_batches
        dev = example_batches[0][0].device
        per_domain = []
        for img, lab, sid in example_batches:
            sid = sid.cpu()
            per_domain.append((img, lab, sid, F.one_hot(sid, num_domains).float().to(dev)))
        fused = _fuse_batches(per_domain, fuse_domains, num_domains, True)
        self.static: List[tuple] = []
        self.slots: List[tuple] = []          # per incoming batch: (static img view, static label view)
        for fb in fused:
            fb = (fb[0].clone(), fb[1].clone(), fb[2], fb[3].clone()) + tuple(fb[4:])
            self.static.append(fb)
            G = fb[4] if len(fb) > 4 else 1
            Bd = fb[0].shape[0] // G
            for g in range(G):
                self.slots.append((fb[0][g * Bd:(g + 1) * Bd], fb[1][g * Bd:(g + 1) * Bd]))
        self.sids = [int(b[2][0]) for b in per_domain]
        ops.enable_device_seed(True)

        def one_step():
            ops.bump_seed()
            return step_fn(self.static)

        if warmup < 1:
            raise ValueError("GraphedStep needs >= 1 eager warm-up step: lazily created state (AdamW moments, workspaces) "
                             "must exist before capture or the graph would re-initialise it on every replay")
        if warmup > 0:                       # eager warm-up steps (they DO update the weights) on a side stream, as torch asks
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    one_step()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = one_step()

    def __call__(self, batches: Sequence[tuple]) -> dict:
        for (s_img, s_lab), b in zip(self.slots, batches):
            s_img.copy_(b[0], non_blocking=True)
            s_lab.copy_(b[1], non_blocking=True)
        assert [int(b[2][0]) for b in batches] == self.sids, "the captured step is specialised to its domain order"
        self.graph.replay()
        # the replay moved the weights (captured optimizer) without touching any host-side tag: the derived-weight caches (W^T,
        # planes, conv layouts) a later EAGER step or eval would hit are one update behind -- invalidate them
        ops.mark_weights_updated()
        return self.out
