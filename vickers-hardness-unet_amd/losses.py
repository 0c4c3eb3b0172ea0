"""Loss modules with the reference's call signatures (train.py:600-601, 438):

    loss_bce  = nn.BCEWithLogitsLoss()            (torch's own; works unchanged on our logits)
    loss_dice = vk.losses.DiceLoss(mode="binary") (this file; replaces smp.losses.DiceLoss)
    loss = loss_bce(logits, y) + loss_dice(logits, y)

``DiceLoss`` and ``BCEDiceLoss`` run the fused HIP reduction (vk_bce_dice_loss); both participate in
autograd.  smp defaults restated: from_logits=True, smooth=0, eps=1e-7, log_loss=False, batch-global
reduction."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from ._lib import VkError, check, lib


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, w_bce, w_dice):
        if not logits.is_cuda:
            raise VkError("loss input is on %s: no CPU fallback in this package" % logits.device)
        x = logits.detach().contiguous().float()
        y = target.detach().contiguous().float().expand_as(x).contiguous()
        sums = torch.empty(8, dtype=torch.float64, device=x.device)
        out = torch.empty(4, dtype=torch.float32, device=x.device)
        need = logits.requires_grad
        dl = torch.empty_like(x) if need else None
        check(lib().vk_bce_dice_loss(x.numel(), x.data_ptr(), y.data_ptr(), sums.data_ptr(), out.data_ptr(),
                                     _lib.ptr(dl), 1.0, float(w_bce), float(w_dice), _lib.current_stream()),
              "vk_bce_dice_loss")
        ctx.dl = dl
        ctx.in_dtype = logits.dtype
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.dl is None:
            return None, None, None, None
        return (ctx.dl * g).to(ctx.in_dtype), None, None, None


class DiceLoss(nn.Module):
    def __init__(self, mode: str = "binary", classes=None, log_loss: bool = False, from_logits: bool = True,
                 smooth: float = 0.0, ignore_index=None, eps: float = 1e-7):
        super().__init__()
        if mode != "binary" or classes is not None or log_loss or not from_logits or smooth != 0.0 \
                or ignore_index is not None or eps != 1e-7:
            raise NotImplementedError("only DiceLoss(mode='binary') with smp defaults is implemented (reference train.py:601)")

    def forward(self, y_pred, y_true):
        return _LossFn.apply(y_pred, y_true, 0.0, 1.0)


class BCEDiceLoss(nn.Module):
    """``BCEWithLogitsLoss()(x, y) + DiceLoss('binary')(x, y)`` in one reduction pass."""

    def forward(self, y_pred, y_true):
        return _LossFn.apply(y_pred, y_true, 1.0, 1.0)
