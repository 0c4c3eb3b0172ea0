"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's training-time augmentation pipeline (train.py:67-113) for
ONE sample with GIVEN random draws, used only by tests/ to check the device kernel (csrc/augment.hip).

The reference composes albumentations transforms (third-party, absent here, un-pinned; they call cv2, also absent):

    A.LongestMaxSize(S, INTER_LINEAR)  A.PadIfNeeded(S, S, BORDER_CONSTANT)                      -> prepost_oracle "train" letterbox
    A.OneOf([HorizontalFlip, VerticalFlip, RandomRotate90], p=0.8)                              -> d4
    A.Rotate(limit=180, border_mode=BORDER_CONSTANT, p=0.6)                                     -> rotate
    A.OneOf([RandomBrightnessContrast, CLAHE(2.0, (8, 8)), GaussianBlur((3, 5))], p=0.8)        -> photo (CLAHE not built)
    A.GaussNoise(p=0.3)                                                                         -> noise
    A.Normalize(ImageNet mean / std)   ToTensorV2()                                             -> float32 CHW

**Parity unpinned**: neither library can be imported and the reference holds no augmented sample.  What is restated, from the
libraries' published behaviour [upstream]:
  * flips / np.rot90: exact pixel permutations.
  * Rotate: cv2.getRotationMatrix2D((w/2 - 0.5, h/2 - 0.5), angle, 1) (positive = counter-clockwise) + warpAffine with a constant
    border of 0, INTER_LINEAR for the image and INTER_NEAREST for the mask.  cv2's warpAffine evaluates the bilinear weights in
    1/32-pixel fixed point; here (oracle and device alike) they are plain float32 in a fixed operation order — a sub-LSB difference
    in a transform whose angle is random anyway.  The mapping, the border rule (taps outside the image contribute the border value)
    and the interpolation modes are cv2's.
  * RandomBrightnessContrast (limits 0.2 / 0.2, brightness_by_max=True): uint8 LUT  clip(v * alpha + beta * 255, 0, 255) truncated.
  * GaussianBlur with sigma 0 and k in {3, 5}: cv2's fixed binomial kernels [1 2 1] / 4 and [1 4 6 4 1] / 16, BORDER_REFLECT_101,
    evaluated exactly in integers with round-half-up.
  * GaussNoise: var ~ U(10, 50), noise ~ N(0, var) per pixel and channel added in float32, clipped, truncated to uint8.  The
    library draws from numpy's generator; here the normal deviate is an Irwin-Hall sum of twelve 16-bit uniforms from a
    counter-based hash (splitmix64 of seed / pixel / channel), so that the device kernel and this file produce the SAME noise field
    bit for bit from one 32-bit seed (variance 1 to 1e-9, support +-6 sigma).
  * Normalize: (v - mean * 255) * (1 / (std * 255)) in float32, as albumentations' functional.normalize computes it.
  * CLAHE is not implemented on the device; the host sampler (vickers-hardness-unet_amd/augment.py) re-normalises the OneOf over the
    two implemented members and says so.
"""
from __future__ import annotations

import numpy as np

F = np.float32
MEAN = np.array([0.485, 0.456, 0.406], dtype=F)
STD = np.array([0.229, 0.224, 0.225], dtype=F)
MEAN255 = (MEAN * F(255.0)).astype(F)
INV_STD255 = (F(1.0) / (STD * F(255.0)).astype(F)).astype(F)

D4_NONE, D4_HFLIP, D4_VFLIP, D4_ROT90_0 = 0, 1, 2, 3          # 3 + k = np.rot90(img, k), k = 0..3
PHOTO_NONE, PHOTO_RBC, PHOTO_CLAHE, PHOTO_BLUR = 0, 1, 2, 3


def apply_d4(a: np.ndarray, d4: int) -> np.ndarray:
    if d4 == D4_HFLIP:
        return a[:, ::-1]
    if d4 == D4_VFLIP:
        return a[::-1]
    if d4 >= D4_ROT90_0:
        return np.rot90(a, d4 - D4_ROT90_0)
    return a


def rotate(img: np.ndarray, mask: np.ndarray, cos_a: float, sin_a: float):
    """warpAffine of a square uint8 image [S][S][3] (bilinear) and mask [S][S] (nearest), constant border 0."""
    S = img.shape[0]
    al, be = F(cos_a), F(sin_a)
    c = F(S / 2.0 - 0.5)
    ys, xs = np.meshgrid(np.arange(S, dtype=F), np.arange(S, dtype=F), indexing="ij")
    dx, dy = (xs - c).astype(F), (ys - c).astype(F)
    sx = (((al * dx).astype(F) - (be * dy).astype(F)).astype(F) + c).astype(F)
    sy = (((be * dx).astype(F) + (al * dy).astype(F)).astype(F) + c).astype(F)
    x0f, y0f = np.floor(sx), np.floor(sy)
    fx, fy = (sx - x0f).astype(F), (sy - y0f).astype(F)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < S) & (xx >= 0) & (xx < S)
        v = img[np.clip(yy, 0, S - 1), np.clip(xx, 0, S - 1)].astype(F)
        return np.where(ok[..., None], v, F(0))

    w0x, w0y = (F(1) - fx).astype(F)[..., None], (F(1) - fy).astype(F)[..., None]
    w1x, w1y = fx[..., None], fy[..., None]
    top = ((tap(y0, x0) * w0x).astype(F) + (tap(y0, x0 + 1) * w1x).astype(F)).astype(F)
    bot = ((tap(y0 + 1, x0) * w0x).astype(F) + (tap(y0 + 1, x0 + 1) * w1x).astype(F)).astype(F)
    val = ((top * w0y).astype(F) + (bot * w1y).astype(F)).astype(F)
    out = np.clip(np.rint(val), 0, 255).astype(np.uint8)
    xi, yi = np.floor((sx + F(0.5)).astype(F)).astype(np.int64), np.floor((sy + F(0.5)).astype(F)).astype(np.int64)
    okm = (yi >= 0) & (yi < S) & (xi >= 0) & (xi < S)
    m = np.where(okm, mask[np.clip(yi, 0, S - 1), np.clip(xi, 0, S - 1)], 0).astype(np.uint8)
    return out, m


def brightness_contrast(img: np.ndarray, alpha: float, beta: float) -> np.ndarray:
    lut = np.arange(256, dtype=F)
    lut = (lut * F(alpha)).astype(F)
    lut = (lut + (F(beta) * F(255.0)).astype(F)).astype(F)
    lut = np.clip(lut, 0, 255).astype(np.uint8)          # astype: truncation, as the library's LUT
    return lut[img]


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def gaussian_blur(img: np.ndarray, k: int) -> np.ndarray:
    w = {3: np.array([1, 2, 1]), 5: np.array([1, 4, 6, 4, 1])}[k]
    r, S = k // 2, img.shape[0]
    idx = np.arange(S)
    a = img.astype(np.int64)
    acc = np.zeros_like(a)
    for i, wy in enumerate(w):
        rows = a[_reflect101(idx + i - r, S)]
        for j, wx in enumerate(w):
            acc += wy * wx * rows[:, _reflect101(idx + j - r, S)]
    tot = int(w.sum()) ** 2
    return ((acc + tot // 2) // tot).astype(np.uint8)      # exact rational, round half up


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return (z ^ (z >> np.uint64(31))).astype(np.uint64)


def noise_isum(seed: int, S: int) -> np.ndarray:
    """Irwin-Hall integer sum of twelve 16-bit uniforms for every (pixel, channel): int64 [S][S][3] in [0, 12 * 65535]."""
    with np.errstate(over="ignore"):
        cnt = (np.arange(S * S * 3, dtype=np.uint64)).reshape(S, S, 3)
        base = (np.uint64(seed) << np.uint64(32)) | cnt
        tot = np.zeros((S, S, 3), dtype=np.int64)
        for j in range(3):
            z = _splitmix64(base * np.uint64(3) + np.uint64(j))
            for sh in (0, 16, 32, 48):
                tot += ((z >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    return tot


def gauss_noise(img: np.ndarray, noise_scale: float, seed: int) -> np.ndarray:
    g = ((noise_isum(seed, img.shape[0]) - 393210).astype(F) * F(noise_scale)).astype(F)
    return np.clip((img.astype(F) + g).astype(F), 0, 255).astype(np.uint8)


def normalize_chw(img: np.ndarray) -> np.ndarray:
    x = ((img.astype(F) - MEAN255).astype(F) * INV_STD255).astype(F)
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1)))


def augment(img_rgb: np.ndarray, mask01: np.ndarray, p: dict):
    """img_rgb uint8 [S][S][3] (letterboxed, RGB), mask01 uint8 [S][S] in {0,1}; p: the draws (keys of vk_aug_params).
    Returns (x float32 [3][S][S], y float32 [1][S][S])."""
    img, m = apply_d4(img_rgb, p["d4"]), apply_d4(mask01, p["d4"])
    if p["rotate"]:
        img, m = rotate(np.ascontiguousarray(img), np.ascontiguousarray(m), p["cos_a"], p["sin_a"])
    if p["photo"] == PHOTO_RBC:
        img = brightness_contrast(img, p["alpha"], p["beta"])
    elif p["photo"] == PHOTO_BLUR:
        img = gaussian_blur(np.ascontiguousarray(img), p["blur_ksize"])
    elif p["photo"] == PHOTO_CLAHE:
        raise NotImplementedError("CLAHE is not part of the device pipeline")
    if p["noise_scale"] > 0:
        img = gauss_noise(np.ascontiguousarray(img), p["noise_scale"], p["noise_seed"])
    return normalize_chw(np.ascontiguousarray(img)), m.astype(F)[None]
