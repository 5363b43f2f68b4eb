"""`OadLoss` ("NONUNIFORM") behind the reference's CRITERIONS registry (criterions/loss.py:6-37): last frame only,
target L2-normalised per row (F.normalize, eps 1e-12), mean over the batch of sum_k -(y_k) log_softmax(logits)_k.
Loss value and dloss/dlogits come from one HIP kernel (csrc/train.hip::oad_loss_kernel)."""
from __future__ import annotations

import torch.nn as nn

from .autograd import oad_loss_autograd
from .registry import CRITERIONS


@CRITERIONS.register("NONUNIFORM")
class OadLoss(nn.Module):
    def __init__(self, cfg, reduction="mean"):
        super().__init__()
        if reduction not in ("mean", "sum"):         # loss.py:30-33 knows these two (anything else leaves `loss` unbound there)
            raise ValueError(f"OadLoss: reduction {reduction!r}: expected 'mean' or 'sum'")
        self.reduction = reduction
        self.num_classes = cfg["num_classes"]

    def forward(self, out_dict, target):
        return oad_loss_autograd(out_dict["logits"], target, self.reduction)
