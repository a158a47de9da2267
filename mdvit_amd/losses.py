"""Step losses of the MDViT train loop as one fused HIP pass over the logits.

Replaces, per domain (multi_train_MDViT.py:147-169 with criterion = [nn.BCELoss(), dice_loss],
Utils/losses.py:8-16, KT_loss = dice_loss):
    output, aux_out = sigmoid(output), sigmoid(aux_out)
    loss     = BCE(output, label)  + dice(output, label)
    aux_loss = BCE(aux_out, label) + dice(aux_out, label)
    kt_loss  = dice(aux_out, output)
"""
from __future__ import annotations

from . import ops


def domain_losses(out_logits, aux_logits, label):
    """-> (loss, aux_loss, kt_loss) 0-dim tensors, differentiable w.r.t. both logit tensors."""
    return ops.seg_losses(out_logits, aux_logits, label)


def seg_loss(out_logits, label):
    """BASE's criterion (multi_train_BASE.py:170-176): BCE(sigmoid(out), y) + dice(sigmoid(out), y)."""
    return ops.seg_losses(out_logits, None, label)[0]
