"""CPU restatement (TEST INFRASTRUCTURE ONLY -- never imported by mdvit_amd) of the two host-side pieces around the hot
path that SURVEY 8f-3 moves onto the device.

PARITY UNPINNED: the arithmetic restated here lives in two third-party packages (medpy, torchvision) that are neither under
/root/reference nor installed in this image, and the reference holds no test vectors for them -- so this file can only be checked
against the packages' published definitions (below) and hand-computed known answers (tests/test_oracle_golden.py:
`test_metric_and_loader_restatements_known_answers`), not against outputs of the packages themselves.  The integer counts behind the metrics are exact.

* Metrics: multi_train_MDViT.py:172-179,275-288 call medpy.metric.binary.dc / jc on `sigmoid(output) > 0.5` and the
  label.  medpy (requirements: medpy, unpinned; algorithm as published in medpy 0.4.0 metric/binary.py) is not under
  /root/reference and not installed here, so its algorithm is restated:
      dc = 2 |A & B| / (|A| + |B|)   (0.0 when the denominator is 0),   jc = |A & B| / |A | B|
  on boolean arrays (`numpy.atleast_1d(x.astype(bool))`).  medpy's jc raises ZeroDivisionError on two empty masks;
  this restatement (and the kernel) return 0.0 there.
* Loader arithmetic: Datasets/create_dataset.py:25-26 (norm01 = clip(x,0,255)/255 in float64), :165-172 (`.float()`,
  permute(2,0,1), torchvision transforms.Normalize(mean, std) = tensor.sub_(mean).div_(std) in fp32).  torchvision is
  not installed; Normalize's two in-place ops are restated.
"""
from __future__ import annotations

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # create_dataset.py:143
IMAGENET_STD = (0.229, 0.224, 0.225)       # create_dataset.py:144


def dc(result: np.ndarray, reference: np.ndarray) -> float:
    result = np.atleast_1d(result.astype(bool)); reference = np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    size = np.count_nonzero(result) + np.count_nonzero(reference)
    return 2.0 * inter / float(size) if size > 0 else 0.0


def jc(result: np.ndarray, reference: np.ndarray) -> float:
    result = np.atleast_1d(result.astype(bool)); reference = np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    union = np.count_nonzero(result | reference)
    return float(inter) / float(union) if union > 0 else 0.0


def train_metrics(out_logits: torch.Tensor, label: torch.Tensor):
    """multi_train_MDViT.py:148,172-177: output = sigmoid(output); (output.cpu().numpy() > 0.5) vs label.cpu().numpy()."""
    o = torch.sigmoid(out_logits.float()).cpu().numpy() > 0.5
    y = label.cpu().numpy()
    return dc(o, y), jc(o, y)


def load_image(img_u8_hwc: np.ndarray) -> torch.Tensor:
    """uint8 [H,W,3] -> fp32 [3,H,W], create_dataset.py:165-172."""
    x = np.clip(img_u8_hwc, 0, 255) / 255                       # norm01, float64
    t = torch.from_numpy(x).float().permute(2, 0, 1).contiguous()
    mean = torch.as_tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.as_tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub_(mean).div_(std)
