"""Loss modules on MI355X — drop-in for the reference's models/losses.py (same class names,
constructor arguments and forward signatures); each forward/backward is a streaming HIP kernel
(idee_bce_* / idee_anomaly_l1_*)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


def _tokens_last(z_q):
    """(N,V,C,T,H,W) -> (N,V,T,H,W,C) without a copy when it is the view VQ_model.forward returns."""
    t = z_q.permute(0, 1, 3, 4, 5, 2)
    return t if t.is_contiguous() else t.contiguous()


class BCE_loss_synthetic(nn.Module):
    """Class-frequency-weighted BCE with logits (losses.py:98-124).  `counts` (2,) overrides the
    local histc, for data-parallel runs that all-reduce the class counts first."""

    def forward(self, pred, target, counts=None):
        target = target.float()
        if counts is None:
            counts = ops.bce_class_counts(target)
        return ops.WeightedBCE.apply(pred.reshape(1, -1), target.reshape(-1), None, counts)[0]


class BCE_loss(nn.Module):
    """losses.py:64-95: weights from the valid pixels' histogram, zero weight on invalid pixels,
    normalised by the number of valid pixels."""

    def forward(self, pred, target, mask_valid, counts=None):
        target, mask_valid = target.float(), mask_valid.float()
        if counts is None:
            counts = ops.bce_class_counts(target, mask_valid)
        return ops.WeightedBCE.apply(pred.reshape(1, -1), target.reshape(-1), mask_valid.reshape(-1), counts)[0]


class Anomaly_L1_loss_synthetic(nn.Module):
    """losses.py:127-168.  pred (N,V,C,T,H,W), mask_extreme (N,H,W), vq_0 (1,C)."""

    def __init__(self, n_dynamic: int = 3, delta_t: int = 8, dim: int = 24):
        super().__init__()
        self.n_dynamic, self.delta_t, self.dim = n_dynamic, delta_t, dim

    def forward(self, pred, mask_extreme, vq_0):
        return ops.AnomalyL1.apply(_tokens_last(pred), mask_extreme.float(), None, vq_0.detach())[0]


class Anomaly_L1_loss(nn.Module):
    """losses.py:15-61.  mask_valid (N,T,H,W) is added to the extreme mask and clipped to 1."""

    def __init__(self, n_dynamic: int = 6, delta_t: int = 8, dim: int = 24):
        super().__init__()
        self.n_dynamic, self.delta_t, self.dim = n_dynamic, delta_t, dim

    def forward(self, pred, mask_extreme, mask_valid, vq_0):
        return ops.AnomalyL1.apply(_tokens_last(pred), mask_extreme.float(), mask_valid.float(), vq_0.detach())[0]
