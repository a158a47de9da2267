"""Synthetic batches with the contract of the reference's loader (Datasets/create_dataset.py:119-189,
SURVEY.md 8d): image = ImageNet-normalised uint8-uniform RGB (B,3,S,S) f32, label = binary filled
ellipse (B,1,S,S) f32, set_id = domain index (B,) int64; one domain per batch."""
from __future__ import annotations

from typing import List, Tuple

import torch

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def make_domain_batch(B: int, S: int, domain: int, seed: int, device="cpu") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g = torch.Generator().manual_seed(seed * 7919 + domain)
    u8 = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    if torch.device(device).type == "cuda":
        # the loader's norm01 + permute + Normalize as ONE kernel on the uint8 image (3 bytes/pixel cross PCIe, not 12)
        from . import ops
        img = ops.image_normalize_u8(u8.to(device))
    else:
        img = u8.float().div_(255.0)
        mean, std = torch.tensor(_MEAN), torch.tensor(_STD)
        img = ((img - mean) / std).permute(0, 3, 1, 2).contiguous()
    r = torch.rand((B, 4), generator=g)
    cy, cx = (0.3 + 0.4 * r[:, 0]) * S, (0.3 + 0.4 * r[:, 1]) * S
    ry, rx = (0.1 + 0.25 * r[:, 2]) * S, (0.1 + 0.25 * r[:, 3]) * S
    yy = torch.arange(S).view(1, S, 1).float()
    xx = torch.arange(S).view(1, 1, S).float()
    lab = ((((yy - cy.view(B, 1, 1)) / ry.view(B, 1, 1)) ** 2 + ((xx - cx.view(B, 1, 1)) / rx.view(B, 1, 1)) ** 2) <= 1.0)
    lab = lab.float().view(B, 1, S, S)
    set_id = torch.full((B,), domain, dtype=torch.long)      # stays on the host, as batch['set_id'] from the DataLoader does
    return img.to(device), lab.to(device), set_id


def make_step_batches(B: int, S: int, rank: int = 0, step: int = 0, device="cpu", domains=(0, 1, 2, 3)) -> List[tuple]:
    """One optimisation step's input: a batch per domain, in the fixed order isic2018, PH2, DMF, SKD = 0..3."""
    return [make_domain_batch(B, S, d, 1234 + rank + 1000 * step, device) for d in domains]
