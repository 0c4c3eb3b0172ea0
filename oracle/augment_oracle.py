"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's training-time augmentation pipeline (train.py:67-113) for
ONE sample with GIVEN random draws, used only by tests/ to check the device kernel (csrc/augment.hip).

The reference composes albumentations transforms (third-party, absent here, un-pinned; they call cv2, also absent):

    A.LongestMaxSize(S, INTER_LINEAR)  A.PadIfNeeded(S, S, BORDER_CONSTANT)                      -> prepost_oracle "train" letterbox
    A.OneOf([HorizontalFlip, VerticalFlip, RandomRotate90], p=0.8)                              -> d4
    A.Rotate(limit=180, border_mode=BORDER_CONSTANT, p=0.6)                                     -> rotate
    A.OneOf([RandomBrightnessContrast, CLAHE(2.0, (8, 8)), GaussianBlur((3, 5))], p=0.8)        -> photo
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
  * CLAHE: OpenCV's algorithm on the L channel of an integer 8-bit RGB <-> L*a*b* conversion (section at the end of this file).
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
        img = clahe_rgb(np.ascontiguousarray(img), clahe_limit(p["clahe_clip"], img.shape[0]))
    if p["noise_scale"] > 0:
        img = gauss_noise(np.ascontiguousarray(img), p["noise_scale"], p["noise_seed"])
    return normalize_chw(np.ascontiguousarray(img)), m.astype(F)[None]


# ------------------------------------------------------------------------------------------------ CLAHE (train.py:98)
# A.CLAHE(clip_limit=2.0, tile_grid_size=(8, 8)) [upstream]: clip ~ U(1, 2); RGB -> LAB (cv2, 8-bit), cv2.createCLAHE(clip, (8, 8)).apply(L),
# LAB -> RGB.  Restated with:
#   * the CLAHE algorithm of OpenCV's clahe.cpp: per-tile 256-bin histogram, clipLimit = max(1, int(clip * tileArea / 256)), excess
#     redistributed as `clipped / 256` to every bin plus the residual to every `256 / residual`-th bin, lut = cvRound(cdf * 255 / tileArea)
#     in float32, bilinear interpolation of the four neighbouring tile LUTs with tile coordinates x / tileW - 0.5 (float32, fixed order);
#   * an 8-bit RGB <-> CIE L*a*b* conversion (sRGB companding, D65) in INTEGER fixed point through three tables built here with numpy
#     (and handed to the device kernel as data), so that device and oracle agree bit for bit.  OpenCV's own 8-bit path uses different
#     fixed-point tables: its L can differ by one level — parity unpinned, as everything cv2 does in this file.
def color_tables():
    """(LIN uint16 [256]: sRGB decode x 4096;  FT int32 [4097]: the Lab f() of t / 4096, x 32768;  ENC uint8 [4097]: sRGB encode of lin / 4096)"""
    v = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)
    LIN = np.rint(lin * 4096.0).astype(np.uint16)
    t = np.arange(4097, dtype=np.float64) / 4096.0
    f = np.where(t > 216.0 / 24389.0, np.cbrt(t), (24389.0 / 27.0 * t + 16.0) / 116.0)
    FT = np.rint(f * 32768.0).astype(np.int32)
    enc = np.where(t <= 0.0031308, 12.92 * t, 1.055 * t ** (1.0 / 2.4) - 0.055)
    ENC = np.clip(np.rint(enc * 255.0), 0, 255).astype(np.uint8)
    return LIN, FT, ENC


# rows: X / Xn, Y, Z / Zn of linear sRGB (D65), x 4096, every row summing to exactly 4096 (white maps to t = 4096)
_M_FWD = np.array([[1777, 1541, 778], [871, 2929, 296], [73, 448, 3575]], dtype=np.int64)
# inverse: linear R, G, B from (X / Xn, Y, Z / Zn), x 4096 (rows sum to 4096)
_M_INV = np.array([[12615, -6296, -2223], [-3773, 7684, 185], [217, -836, 4715]], dtype=np.int64)


def rgb_to_lab_u8(img: np.ndarray, tabs) -> np.ndarray:
    LIN, FT, _ = tabs
    lin = LIN[img].astype(np.int64)                                        # [S][S][3]
    t = (lin @ _M_FWD.T + 2048) >> 12
    t = np.clip(t, 0, 4096)
    f = FT[t].astype(np.int64)
    fx, fy, fz = f[..., 0], f[..., 1], f[..., 2]
    L = (116 * fy * 255 - 16 * 255 * 32768 + 50 * 32768) // (100 * 32768)
    a = ((500 * (fx - fy) + 16384) >> 15) + 128
    b = ((200 * (fy - fz) + 16384) >> 15) + 128
    return np.stack([np.clip(L, 0, 255), np.clip(a, 0, 255), np.clip(b, 0, 255)], axis=-1).astype(np.uint8)


def lab_to_rgb_u8(lab: np.ndarray, tabs) -> np.ndarray:
    _, _, ENC = tabs
    L, a, b = (lab[..., i].astype(np.int64) for i in range(3))
    fy = ((L * 100 * 32768 + 127) // 255 + 16 * 32768 + 58) // 116
    fx = fy + ((a - 128) * 32768 + 250) // 500                                # floor division, also for negatives
    fz = fy - ((b - 128) * 32768 + 100) // 200

    def finv(f):                                                              # t x 4096 from f x 32768
        cube = (f * f * f + (1 << 32)) >> 33
        low = ((f * 116 - 16 * 32768) * 27 * 4096 + (24389 * 32768) // 2) // (24389 * 32768)
        return np.clip(np.where(f > 6780, cube, low), 0, 8192)                # 6 / 29 x 32768 = 6779.6

    t = np.stack([finv(fx), finv(fy), finv(fz)], axis=-1)
    lin = np.clip((t @ _M_INV.T + 2048) >> 12, 0, 4096)
    return ENC[lin]


def clahe_limit(clip: float, S: int) -> int:
    ts = S // 8
    return max(1, int(clip * (ts * ts) / 256.0))                              # clahe.cpp: static_cast<int>(clipLimit * tileSizeTotal / histSize)


def clahe_u8(plane: np.ndarray, limit: int) -> np.ndarray:
    """OpenCV CLAHE on a uint8 [S][S] plane, 8 x 8 tiles (S % 8 == 0), integer clip limit already derived."""
    S = plane.shape[0]
    ts = S // 8
    luts = np.zeros((8, 8, 256), dtype=np.uint8)
    scale = F(255.0) / F(ts * ts)
    for ty in range(8):
        for tx in range(8):
            h = np.bincount(plane[ty * ts:(ty + 1) * ts, tx * ts:(tx + 1) * ts].ravel(), minlength=256).astype(np.int64)
            clipped = int(np.maximum(h - limit, 0).sum())
            h = np.minimum(h, limit)
            batch, residual = clipped // 256, clipped % 256
            h += batch
            if residual:
                step = max(256 // residual, 1)
                idx = np.arange(0, 256, step)[:residual]
                h[idx] += 1
            cdf = np.cumsum(h)
            luts[ty, tx] = np.clip(np.rint((cdf.astype(F) * scale).astype(F)), 0, 255).astype(np.uint8)
    inv = F(1.0) / F(ts)
    c = np.arange(S, dtype=F)
    tf = ((c * inv).astype(F) - F(0.5)).astype(F)
    t1f = np.floor(tf)
    wa = (tf - t1f).astype(F)
    wa1 = (F(1.0) - wa).astype(F)
    t1 = t1f.astype(np.int64)
    t2 = np.minimum(t1 + 1, 7)
    t1 = np.maximum(t1, 0)
    v = plane.astype(np.int64)
    yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    l11 = luts[t1[yy], t1[xx], v].astype(F)
    l12 = luts[t1[yy], t2[xx], v].astype(F)
    l21 = luts[t2[yy], t1[xx], v].astype(F)
    l22 = luts[t2[yy], t2[xx], v].astype(F)
    xa, xa1, ya, ya1 = wa[xx], wa1[xx], wa[yy], wa1[yy]
    top = ((l11 * xa1).astype(F) + (l12 * xa).astype(F)).astype(F)
    bot = ((l21 * xa1).astype(F) + (l22 * xa).astype(F)).astype(F)
    res = ((top * ya1).astype(F) + (bot * ya).astype(F)).astype(F)
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def clahe_rgb(img: np.ndarray, limit: int, tabs=None) -> np.ndarray:
    tabs = tabs or color_tables()
    lab = rgb_to_lab_u8(img, tabs)
    lab[..., 0] = clahe_u8(np.ascontiguousarray(lab[..., 0]), limit)
    return lab_to_rgb_u8(lab, tabs)
