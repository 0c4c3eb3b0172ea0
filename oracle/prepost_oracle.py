"""CPU restatement (numpy) of the steps either side of ``model(x)`` in the reference's inference wrappers
(SURVEY.md §8(f) rank 1).  TEST INFRASTRUCTURE ONLY: imported by tests/ (incl. tests/diag/prepost_bench.py)'s CPU
leg as the checker, never by the product path.

PARITY UNPINNED.  The arithmetic lives in OpenCV (``cv2.resize``), which is not importable in this image,
and the reference holds no fixture of a pre-processed tensor or a post-processed mask.  The functions below
restate the published algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp`` (generic C++ path):

  * INTER_LINEAR, 8-bit: coordinates ``fx = (float)((dx + 0.5) * scale - 0.5)`` with ``scale = 1 / (dst / src)``
    in double, ``sx = floor(fx)``, border clamps (``sx < 0`` -> 0 with weight 0; ``sx >= src-1`` -> src-1 with
    weight 0), weights as 11-bit fixed point ``short(round_half_even(w * 2048))`` for ``1.f - fx`` and ``fx``
    separately, horizontal pass in int32, vertical pass
    ``uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)``.
  * INTER_LINEAR, float32: same coordinates, float weights, ``S[sx]*a0 + S[sx+1]*a1`` then ``S0*b0 + S1*b1``
    in float32 without fused multiply-add (OpenCV's SIMD builds may fuse: compare with a tolerance).
  * INTER_NEAREST: ``sx = min(floor(dx * (1 / (dst / src))), src - 1)``.

What IS pinned: the conventions around the resize — scale rule, rounding of the new size, where the padding
goes, channel order, normalisation constants, thresholds — are restated from the reference's own lines, cited
on each function."""
from __future__ import annotations

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)      # infer_pth_gui.py:14, ui_infer_quadrilateral.py (IMAGENET_MEAN)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)       # infer_pth_gui.py:15


# ------------------------------------------------------------------------------------------------ cv2.resize
def _linear_coords(dst: int, src: int):
    """(index of the left/top sample, float32 weight of the right/bottom sample) for every destination index."""
    inv_scale = float(dst) / float(src)          # cv::resize: inv_scale_x = (double)dsize.width / ssize.width
    scale = 1.0 / inv_scale                      # hal::resize: scale_x = 1. / inv_scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    return s, f


def _round_half_even_to_short(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int32)     # saturate_cast<short>(float) = cvRound


def resize_linear_u8(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """cv2.resize(img, (nw, nh), interpolation=cv2.INTER_LINEAR) for uint8 [h, w] or [h, w, c]."""
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    if (h, w) == (nh, nw):
        return img.copy()
    sx, fx = _linear_coords(nw, w)
    sy, fy = _linear_coords(nh, h)
    a0 = _round_half_even_to_short((np.float32(1.0) - fx) * np.float32(2048.0))
    a1 = _round_half_even_to_short(fx * np.float32(2048.0))
    b0 = _round_half_even_to_short((np.float32(1.0) - fy) * np.float32(2048.0))
    b1 = _round_half_even_to_short(fy * np.float32(2048.0))
    sx1 = np.minimum(sx + 1, w - 1)
    sy1 = np.minimum(sy + 1, h - 1)
    src = img.astype(np.int32).reshape(h, w, -1)
    hres = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]      # [h, nw, c] int32, scale 2048
    r0, r1 = hres[sy], hres[sy1]
    out = ((((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out.reshape((nh, nw) + img.shape[2:])


def resize_linear_f32(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """cv2.resize(img, (nw, nh), interpolation=cv2.INTER_LINEAR) for float32 [h, w]."""
    assert img.dtype == np.float32 and img.ndim == 2
    h, w = img.shape
    if (h, w) == (nh, nw):
        return img.copy()
    sx, fx = _linear_coords(nw, w)
    sy, fy = _linear_coords(nh, h)
    a0, a1 = (np.float32(1.0) - fx).astype(np.float32), fx
    b0, b1 = (np.float32(1.0) - fy).astype(np.float32), fy
    sx1 = np.minimum(sx + 1, w - 1)
    sy1 = np.minimum(sy + 1, h - 1)
    hres = (img[:, sx] * a0[None, :]).astype(np.float32) + (img[:, sx1] * a1[None, :]).astype(np.float32)
    out = (hres[sy] * b0[:, None]).astype(np.float32) + (hres[sy1] * b1[:, None]).astype(np.float32)
    return out.astype(np.float32)


def resize_nearest(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """cv2.resize(img, (nw, nh), interpolation=cv2.INTER_NEAREST) for [h, w]."""
    h, w = img.shape[:2]
    if (h, w) == (nh, nw):
        return img.copy()
    ifx = 1.0 / (float(nw) / float(w))
    ify = 1.0 / (float(nh) / float(h))
    sx = np.minimum(np.floor(np.arange(nw, dtype=np.float64) * ifx).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(nh, dtype=np.float64) * ify).astype(np.int64), h - 1)
    return img[sy][:, sx]


# ------------------------------------------------------------------------------------------------ geometry
def _pyround(v: float) -> int:
    return int(round(v))        # the reference uses Python's round() (half to even) in every wrapper


def geometry_pad_br(h: int, w: int, size: int):
    """infer_pth_gui.py:17-24 letterbox_pad: scale = min(size/h, size/w) (may enlarge), image in the top-left corner,
    zeros right and below.  Returns (scale, nh, nw, top, left)."""
    scale = min(size / h, size / w)
    return scale, _pyround(h * scale), _pyround(w * scale), 0, 0


def geometry_centered(h: int, w: int, size: int):
    """ui_infer_quadrilateral.py:197-216 / ui_infer_rectangle.py:225-245 letterbox_square: scale = min(size/max(h,w), 1)
    (never enlarges), image centred (top = (size-nh)//2, left = (size-nw)//2).  Returns (scale, nh, nw, top, left)."""
    scale = min(size / max(h, w), 1.0)
    nh, nw = _pyround(h * scale), _pyround(w * scale)
    return scale, nh, nw, (size - nh) // 2, (size - nw) // 2


def geometry_train(h: int, w: int, size: int):
    """train.py:70-75 (training) and the validation pipeline below it: A.LongestMaxSize(max_size=size, INTER_LINEAR) — [upstream]
    albumentations scales the LONGEST side to exactly `size` (enlarging small images), new sizes py3round(dim * scale) — then
    A.PadIfNeeded(size, size, border_mode=BORDER_CONSTANT): [upstream] default position "center", pad_top = int((size - nh) / 2.0),
    fill value 0.  (The comment at train.py:72 says right/bottom; the library default is what runs.)  Returns (scale, nh, nw, top, left)."""
    scale = size / max(h, w)
    nh, nw = min(size, _pyround(h * scale)), min(size, _pyround(w * scale))
    return scale, nh, nw, int((size - nh) / 2.0), int((size - nw) / 2.0)


GEOMETRY = {"pad_br": geometry_pad_br, "centered": geometry_centered, "train": geometry_train}


# ------------------------------------------------------------------------------------------------ pre-processing
def letterbox(img_bgr: np.ndarray, size: int, nh: int, nw: int, top: int, left: int, pad_value: int = 0) -> np.ndarray:
    """resize + constant border -> uint8 [size, size, 3] (infer_pth_gui.py:21-23; ui_infer_quadrilateral.py:206-215)."""
    canvas = np.full((size, size, 3), pad_value, dtype=np.uint8)
    canvas[top:top + nh, left:left + nw] = resize_linear_u8(img_bgr, nw, nh)
    return canvas


def normalise_nchw(sq_bgr: np.ndarray) -> np.ndarray:
    """BGR->RGB, /255, (x-mean)/std in float32, HWC->CHW (infer_pth_gui.py:47-49; ui_infer_quadrilateral.py:675-678)."""
    rgb = sq_bgr[:, :, ::-1].astype(np.float32) / np.float32(255.0)
    rgb = (rgb - MEAN) / STD
    return np.ascontiguousarray(np.transpose(rgb, (2, 0, 1))).astype(np.float32)


def preprocess(img_bgr: np.ndarray, size: int, convention: str) -> tuple[np.ndarray, tuple]:
    """-> (float32 [3, size, size], (nh, nw, top, left)); convention "pad_br" (infer_pth_gui), "centered" (Qt wrappers) or
    "train" (the dataset pipeline's LongestMaxSize + PadIfNeeded)."""
    h, w = img_bgr.shape[:2]
    geo = GEOMETRY[convention](h, w, size)
    _, nh, nw, top, left = geo
    return normalise_nchw(letterbox(img_bgr, size, nh, nw, top, left)), (nh, nw, top, left)


# ------------------------------------------------------------------------------------------------ post-processing
def sigmoid_f32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def postprocess_mask(logits_sq: np.ndarray, nh: int, nw: int, top: int, left: int, orig_hw, thresh: float = 0.5) -> np.ndarray:
    """infer_pth_gui.py:50-53, 26-29: mask = (sigmoid(logits) >= thresh) * 255 (uint8), crop, INTER_NEAREST to the original size."""
    mask_sq = (sigmoid_f32(logits_sq) >= np.float32(thresh)).astype(np.uint8) * np.uint8(255)
    crop = mask_sq[top:top + nh, left:left + nw]
    return resize_nearest(crop, orig_hw[1], orig_hw[0])


def postprocess_prob(logits_sq: np.ndarray, nh: int, nw: int, top: int, left: int, orig_hw) -> np.ndarray:
    """ui_infer_quadrilateral.py:705-711, 219-231: prob = sigmoid(logits), crop, INTER_LINEAR to the original size unless
    the crop already has it, clip to [0, 1]."""
    prob = sigmoid_f32(logits_sq)
    crop = np.ascontiguousarray(prob[top:top + nh, left:left + nw])
    if crop.shape != tuple(orig_hw):
        crop = resize_linear_f32(crop, orig_hw[1], orig_hw[0])
    return np.clip(crop, 0.0, 1.0).astype(np.float32)
