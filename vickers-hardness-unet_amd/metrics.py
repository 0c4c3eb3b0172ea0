"""Validation metrics with the reference's exact definitions (train.py:230-255 ``dice_coef``,
:259-281 ``iou_coef``): threshold 0.5 on probabilities, per-image ratio with eps=1e-7, batch mean.
Plain torch ops on whatever device the tensors live on (negligible cost; not on the hot path)."""
import torch


def dice_coef(prob: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> float:
    pred = (prob > 0.5).float()
    inter = (pred * target).sum(dim=(1, 2, 3))
    union = pred.sum(dim=(1, 2, 3)) + target.sum(dim=(1, 2, 3))
    return ((2 * inter + eps) / (union + eps)).mean().item()


def iou_coef(prob: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> float:
    pred = (prob > 0.5).float()
    inter = (pred * target).sum(dim=(1, 2, 3))
    union = pred.sum(dim=(1, 2, 3)) + target.sum(dim=(1, 2, 3)) - inter
    return ((inter + eps) / (union + eps)).mean().item()
