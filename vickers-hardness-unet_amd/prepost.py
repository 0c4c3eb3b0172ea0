"""Host-side mirror of the inference wrappers' pre- and post-processing, running on
the device through libvkunet.so:

    reference                                              here
    infer_pth_gui.letterbox_pad / predict_mask ........... letterbox_geometry(..., "pad_br"), predict_mask
    ui_infer_*.letterbox_square / Segmenter.preprocess ... letterbox_geometry(..., "centered"), Segmenter.preprocess
    train.py LongestMaxSize + PadIfNeeded (train/val) .... letterbox_geometry(..., "train")
    infer_pth_gui.unpad_and_resize_mask .................. postprocess_mask
    ui_infer_*.unletterbox / Segmenter.infer ............. postprocess_prob, Segmenter.infer

The reference resizes on the CPU with cv2 and uploads a float tensor; here the uint8 BGR image is uploaded as it
is (a quarter of the bytes) and resize + border + BGR->RGB + normalisation + planar layout are ONE kernel, and
sigmoid + threshold / un-letterbox are one kernel on the logits.  No CPU fallback: without libvkunet.so these raise."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

Geometry = Tuple[float, int, int, int, int]     # scale, nh, nw, top, left


def letterbox_geometry(h: int, w: int, size: int, convention: str = "centered") -> Geometry:
    """Where the resized image sits in the size x size square.

    "pad_br"   infer_pth_gui.py:17-24: scale = min(size/h, size/w) (may enlarge), image top-left, zeros right/below.
    "centered" ui_infer_quadrilateral.py:197-216, ui_infer_rectangle.py:225-245: scale = min(size/max(h,w), 1), centred.
    "train"    train.py:70-75 (and the validation pipeline below it): albumentations LongestMaxSize(max_size=size) — scale =
               size/max(h,w), may enlarge, new size rounded half-to-even — then PadIfNeeded(size, size, BORDER_CONSTANT), whose
               default position is the centre (top = (size-nh)//2, left = (size-nw)//2), value 0."""
    if convention == "pad_br":
        scale = min(size / h, size / w)
        return scale, int(round(h * scale)), int(round(w * scale)), 0, 0
    if convention == "centered":
        scale = min(size / max(h, w), 1.0)
        nh, nw = int(round(h * scale)), int(round(w * scale))
        return scale, nh, nw, (size - nh) // 2, (size - nw) // 2
    if convention == "train":
        scale = size / max(h, w)
        nh, nw = min(size, int(round(h * scale))), min(size, int(round(w * scale)))
        return scale, nh, nw, (size - nh) // 2, (size - nw) // 2
    raise ValueError(f"unknown letterbox convention {convention!r}")


def _desc(h: int, w: int, size: int, geo: Geometry, stride: int = 0, pad_value: int = 0) -> L.vk_letterbox_desc:
    _, nh, nw, top, left = geo
    return L.vk_letterbox_desc(h, w, stride, size, nh, nw, top, left, pad_value)


def _as_device_u8(img_bgr, device) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(img_bgr)) if isinstance(img_bgr, np.ndarray) else img_bgr
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected a uint8 BGR image of shape [h, w, 3]")
    return t.to(device, non_blocking=True).contiguous()


def preprocess(img_bgr, size: int = 512, convention: str = "centered", device=None, pad_value: int = 0,
               out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Tuple]:
    """uint8 BGR [h, w, 3] (numpy or tensor) -> (float32 [1, 3, size, size] on the device, (scale, geometry, (h, w))).
    `out`: optional [3, size, size] float32 slice of a batch tensor to write into."""
    device = torch.device(device if device is not None else "cuda")
    src = _as_device_u8(img_bgr, device)
    h, w = int(src.shape[0]), int(src.shape[1])
    geo = letterbox_geometry(h, w, size, convention)
    x = out if out is not None else torch.empty(3, size, size, dtype=torch.float32, device=device)
    if x.shape != (3, size, size) or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("out must be a contiguous float32 [3, size, size] tensor")
    d = _desc(h, w, size, geo, stride=3 * w, pad_value=pad_value)
    L.check(L.lib().vk_letterbox_preprocess(C.byref(d), src.data_ptr(), x.data_ptr(), L.current_stream()), "vk_letterbox_preprocess")
    return x.unsqueeze(0), (geo[0], geo, (h, w))


def preprocess_batch(images: Sequence, size: int = 512, convention: str = "centered", device=None) -> Tuple[torch.Tensor, List[Tuple]]:
    """A list of differently sized BGR images -> one [n, 3, size, size] input batch (+ per-image metadata)."""
    device = torch.device(device if device is not None else "cuda")
    x = torch.empty(len(images), 3, size, size, dtype=torch.float32, device=device)
    metas = [preprocess(im, size, convention, device, out=x[i])[1] for i, im in enumerate(images)]
    return x, metas


def postprocess_mask(logits_sq: torch.Tensor, meta: Tuple, thresh: float = 0.5) -> torch.Tensor:
    """logits [S, S] -> uint8 {0,255} [h, w] on the device (infer_pth_gui.py:50-53 + unpad_and_resize_mask :26-29)."""
    _, geo, (h, w) = meta
    lg = logits_sq.reshape(logits_sq.shape[-2], logits_sq.shape[-1]).contiguous().float()
    out = torch.empty(h, w, dtype=torch.uint8, device=lg.device)
    d = _desc(h, w, int(lg.shape[-1]), geo)
    L.check(L.lib().vk_letterbox_postprocess_mask(C.byref(d), lg.data_ptr(), float(thresh), out.data_ptr(), L.current_stream()),
            "vk_letterbox_postprocess_mask")
    return out


def postprocess_prob(logits_sq: torch.Tensor, meta: Tuple) -> torch.Tensor:
    """logits [S, S] -> probability map float32 [h, w] in [0, 1] on the device (Segmenter.infer, ui_infer_quadrilateral.py:705-711)."""
    _, geo, (h, w) = meta
    lg = logits_sq.reshape(logits_sq.shape[-2], logits_sq.shape[-1]).contiguous().float()
    out = torch.empty(h, w, dtype=torch.float32, device=lg.device)
    d = _desc(h, w, int(lg.shape[-1]), geo)
    L.check(L.lib().vk_letterbox_postprocess_prob(C.byref(d), lg.data_ptr(), out.data_ptr(), L.current_stream()),
            "vk_letterbox_postprocess_prob")
    return out


def predict_mask(model, bgr: np.ndarray, device=None, img_size: int = 512, thresh: float = 0.5) -> np.ndarray:
    """infer_pth_gui.py:45-53: BGR image -> uint8 mask {0,255} of the original size (model in eval mode, fp32)."""
    device = torch.device(device if device is not None else "cuda")
    x, meta = preprocess(bgr, img_size, "pad_br", device)
    with torch.no_grad():
        logits = model(x)
    return postprocess_mask(logits[0, 0], meta, thresh).cpu().numpy()


class Segmenter:
    """The PyTorch branch of the Qt wrappers' Segmenter (ui_infer_quadrilateral.py:596-711, ui_infer_rectangle.py:455-564):
    `preprocess(img_bgr) -> (inp, meta)` and `infer(img_bgr) -> float32 probability map of the original size`."""

    def __init__(self, model, img_size: int = 512, device=None):
        self.device = torch.device(device if device is not None else "cuda")
        self.model = model.to(self.device).eval()
        self.img_size = img_size

    def preprocess(self, img_bgr: np.ndarray):
        return preprocess(img_bgr, self.img_size, "centered", self.device)

    def infer(self, img_bgr: np.ndarray) -> np.ndarray:
        x, meta = self.preprocess(img_bgr)
        with torch.no_grad():
            logits = self.model(x)
        return postprocess_prob(logits[0, 0], meta).cpu().numpy()

    def infer_batch(self, images: Sequence[np.ndarray]) -> List[np.ndarray]:
        """Several images through ONE forward pass (the reference runs them one by one)."""
        x, metas = preprocess_batch(images, self.img_size, "centered", self.device)
        with torch.no_grad():
            logits = self.model(x)
        return [postprocess_prob(logits[i, 0], m).cpu().numpy() for i, m in enumerate(metas)]
