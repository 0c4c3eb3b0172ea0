"""``train_one_epoch`` / ``validate`` with the reference's signatures and step order
(train.py:381-459, 461-529), minus tqdm and the JPEG dump.  ``loader`` yields ``(x, y, name)`` exactly
like ``VickersDataset`` (train.py:195-200): x fp32 [N,3,S,S] ImageNet-normalised, y fp32 [N,1,S,S]."""
from __future__ import annotations

import numpy as np
import torch

from .metrics import dice_coef, iou_coef


def train_one_epoch(model, loader, optimizer, loss_fn_bce, loss_fn_dice, device, scaler=None):
    model.train()
    t_loss, count = 0.0, 0
    use_amp = (device == "cuda")
    for x, y, _ in loader:
        x, y = x.to(device), y.to(device)
        optimizer.zero_grad(set_to_none=True)
        with torch.amp.autocast(device_type=("cuda" if use_amp else "cpu"),
                                dtype=(torch.float16 if use_amp else torch.bfloat16), enabled=use_amp):
            logits = model(x)
            loss = loss_fn_bce(logits, y) + loss_fn_dice(logits, y)
        if scaler is not None and use_amp:
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            loss.backward()
            optimizer.step()
        t_loss += loss.item() * x.size(0)
        count += x.size(0)
    return t_loss / max(1, count)


@torch.no_grad()
def validate(model, loader, loss_fn_bce, loss_fn_dice, device):
    model.eval()
    v_loss, count = 0.0, 0
    dices, ious = [], []
    for x, y, _ in loader:
        x, y = x.to(device), y.to(device)
        logits = model(x)
        loss = loss_fn_bce(logits, y) + loss_fn_dice(logits, y)
        v_loss += loss.item() * x.size(0)
        count += x.size(0)
        prob = torch.sigmoid(logits)
        dices.append(dice_coef(prob, y))
        ious.append(iou_coef(prob, y))
    return v_loss / max(1, count), float(np.mean(dices)), float(np.mean(ious))
