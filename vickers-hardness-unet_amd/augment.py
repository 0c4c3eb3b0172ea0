"""Device-resident dataset + training-time augmentation: the job of the reference's ``VickersDataset``
(train.py:35-200) — cv2 read, letterbox, albumentations pipeline, Normalize, ToTensorV2 — without a CPU in the loop.

    reference                                                     here
    VickersDataset(aug=True).__getitem__  (train.py:173-200)      DeviceDataset.batch(indices, sampler)  -> x [n,3,S,S], y [n,1,S,S]
    VickersDataset(aug=False)             (train.py:114-130)      DeviceDataset.batch(indices, None)
    A.Compose([...]) random draws          (train.py:67-113)      AugmentSampler (same transforms, probabilities and ranges)

The images are letterboxed once (``vk_letterbox_u8`` / ``vk_letterbox_mask_u8``, geometry "train") and stay in HBM as uint8
(183 micrographs at 512x512 are 144 MB); every step one fused kernel (``vk_augment_batch``) applies the geometric and
photometric transforms of the batch and writes normalised float32 tensors (samples that draw CLAHE take one extra pass for
their tile histograms).  At 2,500 img/s the reference's loader
(``num_workers=0``, cv2 + albumentations on the training thread, train.py:586-589) would be the bottleneck by two to three
orders of magnitude.

``A.CLAHE`` (train.py:98) is OpenCV's algorithm on the L channel of an 8-bit L*a*b* image; the colour conversion here is an
integer fixed-point one through three tables (``color_tables``) so that device and checker agree bit for bit — OpenCV's own
8-bit conversion uses different tables and can differ by a level.  No CPU fallback: without libvkunet.so these raise."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .prepost import letterbox_geometry

D4_NONE, D4_HFLIP, D4_VFLIP, D4_ROT90 = 0, 1, 2, 3
PHOTO_NONE, PHOTO_RBC, PHOTO_CLAHE, PHOTO_BLUR = 0, 1, 2, 3


def color_tables() -> np.ndarray:
    """int32 [VK_AUG_TABLE_INTS] = LIN [256] | FT [4097] | ENC [4097]: sRGB decode x 4096, the Lab f() of t / 4096 x 32768, and the sRGB
    encode of lin / 4096 (0..255) — constants of the fixed-point RGB <-> L*a*b* conversion, built once and uploaded."""
    v = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4)
    t = np.arange(4097, dtype=np.float64) / 4096.0
    f = np.where(t > 216.0 / 24389.0, np.cbrt(t), (24389.0 / 27.0 * t + 16.0) / 116.0)
    enc = np.where(t <= 0.0031308, 12.92 * t, 1.055 * t ** (1.0 / 2.4) - 0.055)
    return np.concatenate([np.rint(lin * 4096.0), np.rint(f * 32768.0), np.clip(np.rint(enc * 255.0), 0, 255)]).astype(np.int32)


def clahe_limit(clip: float, size: int) -> int:
    """OpenCV's integer clip limit for 8 x 8 tiles of a ``size`` x ``size`` image (clahe.cpp: int(clipLimit * tileArea / 256), >= 1)."""
    ts = size // 8
    return max(1, int(clip * (ts * ts) / 256.0))


class AugmentSampler:
    """The random draws of the reference's ``A.Compose`` (train.py:67-113), made on the host exactly where albumentations makes
    them; ``sample()`` returns one ``vk_aug_params``-shaped dict.  [upstream] parameter ranges restated from albumentations'
    defaults: RandomBrightnessContrast limits 0.2 / 0.2 (brightness_by_max), GaussianBlur kernel 3 or 5 with sigma 0,
    GaussNoise var_limit (10, 50), RandomRotate90 factor 0..3, Rotate angle uniform in [-180, 180], CLAHE clip limit uniform in
    (1, clip_limit = 2).  ``clahe``: "on" (the reference's three-member OneOf), "skip" (draw among the other two members) or "none"
    (a CLAHE draw leaves the sample without a photometric transform)."""

    def __init__(self, seed: Optional[int] = None, clahe: str = "on"):
        if clahe not in ("on", "skip", "none"):
            raise ValueError("clahe must be 'on', 'skip' or 'none'")
        self.rng = np.random.default_rng(seed)
        self.clahe = clahe

    def sample(self) -> dict:
        r = self.rng
        p = dict(d4=D4_NONE, rotate=0, cos_a=1.0, sin_a=0.0, photo=PHOTO_NONE, alpha=1.0, beta=0.0, blur_ksize=3,
                 noise_scale=0.0, noise_seed=0, clahe_clip=1.0)
        if r.random() < 0.8:                                    # OneOf([HFlip, VFlip, RandomRotate90], p=0.8), train.py:81-85
            which = int(r.integers(3))
            p["d4"] = (D4_HFLIP, D4_VFLIP, D4_ROT90 + int(r.integers(4)))[which]
        if r.random() < 0.6:                                    # Rotate(limit=180, p=0.6), train.py:89
            ang = math.radians(float(r.uniform(-180.0, 180.0)))
            p["rotate"], p["cos_a"], p["sin_a"] = 1, math.cos(ang), math.sin(ang)
        if r.random() < 0.8:                                    # OneOf([RBC, CLAHE, GaussianBlur], p=0.8), train.py:96-100
            which = (0, 2)[int(r.integers(2))] if self.clahe == "skip" else int(r.integers(3))
            if which == 0:
                p["photo"] = PHOTO_RBC
                p["alpha"] = 1.0 + float(r.uniform(-0.2, 0.2))
                p["beta"] = float(r.uniform(-0.2, 0.2))
            elif which == 1 and self.clahe == "on":
                p["photo"] = PHOTO_CLAHE
                p["clahe_clip"] = float(r.uniform(1.0, 2.0))
            elif which == 2:
                p["photo"] = PHOTO_BLUR
                p["blur_ksize"] = int(r.choice([3, 5]))
        if r.random() < 0.3:                                    # GaussNoise(p=0.3), train.py:104
            sigma = math.sqrt(float(r.uniform(10.0, 50.0)))
            p["noise_scale"] = sigma / 65536.0
            p["noise_seed"] = int(r.integers(0, 2 ** 32))
        return p


IDENTITY = dict(d4=D4_NONE, rotate=0, cos_a=1.0, sin_a=0.0, photo=PHOTO_NONE, alpha=1.0, beta=0.0, blur_ksize=3, noise_scale=0.0,
                noise_seed=0, clahe_clip=1.0)


def _params_array(draws: Sequence[dict], size: int):
    arr = (L.vk_aug_params * len(draws))()
    for i, d in enumerate(draws):
        arr[i] = L.vk_aug_params(int(d["d4"]), int(d["rotate"]), float(d["cos_a"]), float(d["sin_a"]), int(d["photo"]),
                                 float(d["alpha"]), float(d["beta"]), int(d["blur_ksize"]), float(d["noise_scale"]),
                                 int(d["noise_seed"]) & 0xFFFFFFFF, clahe_limit(float(d.get("clahe_clip", 1.0)), size))
    return arr


class DeviceDataset:
    """All (image, mask) pairs letterboxed to ``img_size`` x ``img_size`` and resident on the device as uint8."""

    def __init__(self, images_bgr: Sequence[np.ndarray], masks: Sequence[np.ndarray], img_size: int = 512, device=None,
                 names: Optional[Sequence[str]] = None):
        if len(images_bgr) != len(masks) or not len(images_bgr):
            raise ValueError("need one mask per image")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise L.VkError("DeviceDataset lives on the MI355X; there is no CPU path")
        self.S = int(img_size)
        self.names = list(names) if names is not None else [str(i) for i in range(len(images_bgr))]
        n, S = len(images_bgr), self.S
        self.images = torch.empty(n, S, S, 3, dtype=torch.uint8, device=self.device)
        self.masks = torch.empty(n, S, S, dtype=torch.uint8, device=self.device)
        lib, st = L.lib(), L.current_stream()
        keep = []
        for i, (im, mk) in enumerate(zip(images_bgr, masks)):
            im = np.ascontiguousarray(im)
            mk = np.ascontiguousarray(mk[:, :, 0] if mk.ndim == 3 else mk)           # train.py:163-164
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 or mk.dtype != np.uint8 or mk.shape != im.shape[:2]:
                raise ValueError(f"item {i}: expected uint8 BGR [h, w, 3] and uint8 mask [h, w]")
            h, w = im.shape[:2]
            _, nh, nw, top, left = letterbox_geometry(h, w, S, "train")
            ti, tm = torch.from_numpy(im).to(self.device), torch.from_numpy(mk).to(self.device)
            keep += [ti, tm]
            d = L.vk_letterbox_desc(h, w, 3 * w, S, nh, nw, top, left, 0)
            L.check(lib.vk_letterbox_u8(C.byref(d), ti.data_ptr(), self.images[i].data_ptr(), st), "vk_letterbox_u8")
            d = L.vk_letterbox_desc(h, w, w, S, nh, nw, top, left, 0)
            L.check(lib.vk_letterbox_mask_u8(C.byref(d), tm.data_ptr(), self.masks[i].data_ptr(), st), "vk_letterbox_mask_u8")
        torch.cuda.synchronize(self.device)
        del keep
        self._params_dev = None
        self._tables = torch.from_numpy(color_tables()).to(self.device)
        self._clahe_ws = None

    def __len__(self) -> int:
        return int(self.images.shape[0])

    def batch(self, indices: Sequence[int], sampler: Optional[AugmentSampler] = None, draws: Optional[Sequence[dict]] = None
              ) -> Tuple[torch.Tensor, torch.Tensor, List[str]]:
        """One stream at a time: the parameter scratch and the CLAHE workspace of the dataset are reused by every call, so
        batches requested on different streams must be ordered by the caller (consecutive calls on one stream are).

        ``(x float32 [n,3,S,S], y float32 [n,1,S,S], names)`` for the given items — the tuple ``VickersDataset`` + ``DataLoader``
        hand to the training loop (train.py:195-200, 423).  ``sampler=None`` and ``draws=None``: the validation pipeline (no random
        transforms, train.py:114-130).  ``draws``: explicit parameter dicts (tests)."""
        idx = [int(i) for i in indices]
        n, S = len(idx), self.S
        if draws is None:
            draws = [sampler.sample() if sampler is not None else IDENTITY for _ in idx]
        if len(draws) != n:
            raise ValueError("one set of draws per index")
        arr = _params_array(draws, S)
        if self._params_dev is None or self._params_dev.numel() < n * C.sizeof(L.vk_aug_params):
            self._params_dev = torch.empty(max(n, 64) * C.sizeof(L.vk_aug_params), dtype=torch.uint8, device=self.device)
        index = torch.tensor(idx, dtype=torch.int32, device=self.device)
        x = torch.empty(n, 3, S, S, dtype=torch.float32, device=self.device)
        y = torch.empty(n, 1, S, S, dtype=torch.float32, device=self.device)
        ws_ptr, ws_bytes = None, 0
        if any(int(d["photo"]) == PHOTO_CLAHE for d in draws):
            ws_bytes = int(L.lib().vk_augment_workspace_bytes(n, S))
            if self._clahe_ws is None or self._clahe_ws.numel() < ws_bytes:
                self._clahe_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            ws_ptr = self._clahe_ws.data_ptr()
        L.check(L.lib().vk_augment_batch(n, S, len(self), self.images.data_ptr(), self.masks.data_ptr(), index.data_ptr(), arr,
                                         self._params_dev.data_ptr(), self._tables.data_ptr(), ws_ptr, ws_bytes, x.data_ptr(),
                                         y.data_ptr(), L.current_stream()),
                "vk_augment_batch")
        return x, y, [self.names[i] if 0 <= i < len(self.names) else "?" for i in idx]

    def loader(self, batch_size: int, shuffle: bool = True, sampler: Optional[AugmentSampler] = None, seed: Optional[int] = None):
        """Iterate ``(x, y, names)`` over one epoch like ``DataLoader(VickersDataset(...), batch_size, shuffle)`` (train.py:586-589)."""
        order = np.arange(len(self))
        if shuffle:
            np.random.default_rng(seed).shuffle(order)
        for b in range(0, len(order), batch_size):
            yield self.batch(order[b:b + batch_size].tolist(), sampler)
