"""Validation metrics on the device (reference: ``dice_coef`` train.py:230-255, ``iou_coef`` :259-281, used by ``validate``
train.py:518-522).  One HIP pass (``vk_seg_metrics``) leaves {I, P, T} per image; Dice and IoU come out of a second
single-workgroup launch — no 0/1 prediction tensor, no products, no torch reductions.  CUDA tensors only."""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib
from ._lib import VkError, check, lib


def seg_metrics_device(pred: torch.Tensor, target: torch.Tensor, from_logits: bool = False, threshold: float = 0.5,
                       eps: float = 1e-7) -> torch.Tensor:
    """Device tensor of 2 + 2 N floats: [mean dice, mean iou, dice_0, iou_0, ...]; no host synchronisation."""
    if not (pred.is_cuda and target.is_cuda):
        raise VkError("seg metrics take CUDA tensors (%s / %s): no CPU fallback in this package" % (pred.device, target.device))
    if pred.dim() < 2 or pred.shape[0] < 1:
        raise ValueError("expected [N, ...] predictions, got %s" % (tuple(pred.shape),))
    n = pred.shape[0]
    p = pred.detach().contiguous().float()
    t = target.detach().float().expand_as(p).contiguous()
    per_image = p.numel() // n
    L = lib()
    ws_bytes = L.vk_seg_metrics_workspace_bytes(n)
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=p.device)
    out = torch.empty(2 + 2 * n, dtype=torch.float32, device=p.device)
    check(L.vk_seg_metrics(n, per_image, p.data_ptr(), t.data_ptr(), 1 if from_logits else 0, float(threshold), float(eps),
                           ws.data_ptr(), ws_bytes, out.data_ptr(), _lib.current_stream()), "vk_seg_metrics")
    return out


def seg_metrics(pred: torch.Tensor, target: torch.Tensor, from_logits: bool = False, threshold: float = 0.5,
                eps: float = 1e-7) -> Tuple[float, float]:
    """(batch-mean Dice, batch-mean IoU) as Python floats: one read-back of two values for both metrics."""
    d, u = seg_metrics_device(pred, target, from_logits, threshold, eps)[:2].tolist()
    return d, u


def dice_coef(prob: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> float:
    """Signature of the reference's ``dice_coef(prob, target, eps)``: probabilities in, float out."""
    return seg_metrics(prob, target, False, 0.5, eps)[0]


def iou_coef(prob: torch.Tensor, target: torch.Tensor, eps: float = 1e-7) -> float:
    """Signature of the reference's ``iou_coef(prob, target, eps)``."""
    return seg_metrics(prob, target, False, 0.5, eps)[1]
