"""Host-side mirror of the reference's geometry post-processing, running on the device
through libvkunet.so (``vk_geom_minarearect``, csrc/geometry.hip):

    reference                                                        here
    ui_infer_rectangle.postprocess_minarearect_multi (:291-381) .... postprocess_minarearect_multi (one map, same return value)
                                                                     postprocess_minarearect_batch (B maps of one size, one call)
    ui_infer_quadrilateral.postprocess_minarearect_multi (:423-530)
      + robust_quadrilateral_from_contour and helpers (:262-420) .... postprocess_quadrilateral_multi / postprocess_quadrilateral_batch
                                                                     (``vk_geom_quadrilateral``: the 4-vertex fit of the newer GUI)

The reference thresholds the probability map with numpy and hands everything else to cv2 on the CPU, one image at a time on the
GUI thread; here threshold, 3x3 open/close, 8-connected labelling with the area filter, convex hull, minimum-area rectangle,
int32 corners and the two diagonals are device kernels over the whole batch, and only the few detection records travel back.
No CPU fallback: without libvkunet.so or with a CPU tensor these raise."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib as L

BIN_THRESH = 0.50        # ui_infer_rectangle.py:45 (the quadrilateral GUI uses 0.45, ui_infer_quadrilateral.py:46)
MIN_AREA_FRAC = 0.0008   # ui_infer_rectangle.py:46
MORPH_KERNEL = 3
OPEN_ITER = 1
CLOSE_ITER = 1


def _records(dets, count: int, cap: int) -> List[Dict]:
    out = []
    for i in range(min(count, cap)):
        d = dets[i]
        box = np.array(list(d.box), dtype=np.int32).reshape(4, 2)
        out.append({"label": int(d.label), "area": int(d.area), "box": box, "center": (float(d.cx), float(d.cy)),
                    "d1": float(d.d1), "d2": float(d.d2), "d_mean": float(d.d_mean),
                    "size": (float(d.rw), float(d.rh)), "direction": (float(d.ux), float(d.uy)), "hull_vertices": int(d.hull_n)})
    out.sort(key=lambda r: r["area"], reverse=True)          # ui_infer_rectangle.py:379 (stable: ties keep label order)
    return out


def postprocess_minarearect_batch(prob: torch.Tensor, bin_thresh: float = BIN_THRESH, min_area_frac: float = MIN_AREA_FRAC,
                                  morph_kernel: int = MORPH_KERNEL, open_iter: int = OPEN_ITER, close_iter: int = CLOSE_ITER,
                                  max_components: int = 64) -> Tuple[torch.Tensor, List[List[Dict]]]:
    """``prob``: float32 [B, h, w] probability maps on the device.  Returns (clean uint8 [B, h, w] in {0, 255} on the device,
    per map the reference's detection list).  One host synchronisation at the end (the detection records are read back)."""
    if not isinstance(prob, torch.Tensor) or prob.dim() != 3:
        raise ValueError("expected a float32 tensor [B, h, w]")
    if not prob.is_cuda:
        raise L.VkError("probability maps are on %s: this package runs on an MI355X only and has no CPU fallback" % prob.device)
    prob = prob.contiguous().float()
    B, h, w = (int(v) for v in prob.shape)
    min_area = max(200, int(min_area_frac * h * w))          # ui_infer_rectangle.py:322
    desc = L.vk_geom_desc(h, w, float(bin_thresh), int(morph_kernel), int(open_iter), int(close_iter), min_area, int(max_components))
    lib = L.lib()
    nbytes = lib.vk_geom_workspace_bytes(C.byref(desc), B)
    if nbytes < 0:
        L.check(-1, "vk_geom_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=prob.device)
    clean = torch.empty(B, h, w, dtype=torch.uint8, device=prob.device)
    rec_bytes = C.sizeof(L.vk_geom_det)
    dets = torch.zeros(B * max_components * rec_bytes, dtype=torch.uint8, device=prob.device)
    counts = torch.zeros(B, dtype=torch.int32, device=prob.device)
    L.check(lib.vk_geom_minarearect(C.byref(desc), B, prob.data_ptr(), clean.data_ptr(), dets.data_ptr(), counts.data_ptr(),
                                    ws.data_ptr(), nbytes, L.current_stream()), "vk_geom_minarearect")
    host = dets.cpu().numpy().tobytes()                       # synchronises
    cnt = counts.cpu().tolist()
    arr = (L.vk_geom_det * (B * max_components)).from_buffer_copy(host)
    out = [_records(arr[b * max_components:(b + 1) * max_components], cnt[b], max_components) for b in range(B)]
    return clean, out


def postprocess_minarearect_multi(img_bgr, prob01, bin_thresh: float = BIN_THRESH, min_area_frac: float = MIN_AREA_FRAC,
                                  morph_kernel: int = MORPH_KERNEL, open_iter: int = OPEN_ITER, close_iter: int = CLOSE_ITER,
                                  device=None):
    """Signature and return value of ui_infer_rectangle.postprocess_minarearect_multi (:291-381): ``(clean_bin uint8 [h, w],
    detections)`` with ``detections`` sorted by area, each ``{"label", "area", "box" int32 [4, 2], "center", "d1", "d2", "d_mean"}``
    (+ "size", "direction", "hull_vertices").  ``img_bgr`` is accepted and ignored, as in the reference.  ``prob01`` may be a numpy
    array (uploaded) or a device tensor (what ``vk.Segmenter.infer`` can hand over without a round trip)."""
    if isinstance(prob01, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(prob01, dtype=np.float32)).to(torch.device(device if device is not None else "cuda"))
    else:
        t = prob01
    clean, dets = postprocess_minarearect_batch(t.reshape(1, *t.shape[-2:]), bin_thresh, min_area_frac, morph_kernel, open_iter, close_iter)
    return clean[0].cpu().numpy(), dets[0]


# ------------------------------------------------------------------------------------------------ the newer GUI: 4-vertex fit
BIN_THRESH_QUAD = 0.45   # ui_infer_quadrilateral.py:46
FIT_OUTSET_PX = 2        # ui_infer_quadrilateral.py:433

_BRANCH = {0: "none", 1: "bisection", 2: "subsample", 3: "extremes"}


def _quad_records(dets, count: int, cap: int) -> List[Dict]:
    out = []
    for i in range(min(count, cap)):
        d = dets[i]
        if not d.valid:                                   # `if quad is None: continue` (ui_infer_quadrilateral.py:498-499)
            continue
        box = np.array(list(d.box), dtype=np.int32).reshape(4, 2)
        out.append({"label": int(d.label), "area": int(d.area), "box": box, "center": (float(d.cx), float(d.cy)),
                    "d1": float(d.d1), "d2": float(d.d2), "d_mean": float(d.d_mean),
                    "quality": float(d.quality), "branch": _BRANCH.get(int(d.branch), "?"), "n_candidates": int(d.n_candidates),
                    "contour_points": int(d.contour_n), "hull_vertices": int(d.hull_n), "flags": int(d.flags)})
    out.sort(key=lambda r: r["area"], reverse=True)          # ui_infer_quadrilateral.py:526
    return out


def postprocess_quadrilateral_batch(prob: torch.Tensor, bin_thresh: float = BIN_THRESH_QUAD, min_area_frac: float = MIN_AREA_FRAC,
                                    morph_kernel: int = MORPH_KERNEL, open_iter: int = OPEN_ITER, close_iter: int = CLOSE_ITER,
                                    fit_outset_px: int = FIT_OUTSET_PX, max_components: int = 64) -> Tuple[torch.Tensor, List[List[Dict]]]:
    """``prob``: float32 [B, h, w] on the device -> (clean uint8 [B, h, w] on the device, per map the detection list of
    ui_infer_quadrilateral.postprocess_minarearect_multi).  One host synchronisation (the records are read back)."""
    if not isinstance(prob, torch.Tensor) or prob.dim() != 3:
        raise ValueError("expected a float32 tensor [B, h, w]")
    if not prob.is_cuda:
        raise L.VkError("probability maps are on %s: this package runs on an MI355X only and has no CPU fallback" % prob.device)
    prob = prob.contiguous().float()
    B, h, w = (int(v) for v in prob.shape)
    min_area = max(200, int(min_area_frac * h * w))          # ui_infer_quadrilateral.py:457
    desc = L.vk_geom_desc(h, w, float(bin_thresh), int(morph_kernel), int(open_iter), int(close_iter), min_area, int(max_components))
    lib = L.lib()
    nbytes = lib.vk_geom_workspace_bytes(C.byref(desc), B)
    if nbytes < 0:
        L.check(-1, "vk_geom_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=prob.device)
    clean = torch.empty(B, h, w, dtype=torch.uint8, device=prob.device)
    rec_bytes = C.sizeof(L.vk_geom_quad)
    dets = torch.zeros(B * max_components * rec_bytes, dtype=torch.uint8, device=prob.device)
    counts = torch.zeros(B, dtype=torch.int32, device=prob.device)
    L.check(lib.vk_geom_quadrilateral(C.byref(desc), int(fit_outset_px), B, prob.data_ptr(), clean.data_ptr(), dets.data_ptr(),
                                      counts.data_ptr(), ws.data_ptr(), nbytes, L.current_stream()), "vk_geom_quadrilateral")
    host = dets.cpu().numpy().tobytes()
    cnt = counts.cpu().tolist()
    arr = (L.vk_geom_quad * (B * max_components)).from_buffer_copy(host)
    out = [_quad_records(arr[b * max_components:(b + 1) * max_components], cnt[b], max_components) for b in range(B)]
    return clean, out


def postprocess_quadrilateral_multi(img_bgr, prob01, bin_thresh: float = BIN_THRESH_QUAD, min_area_frac: float = MIN_AREA_FRAC,
                                    morph_kernel: int = MORPH_KERNEL, open_iter: int = OPEN_ITER, close_iter: int = CLOSE_ITER,
                                    fit_outset_px: int = FIT_OUTSET_PX, device=None):
    """Signature and return value of ui_infer_quadrilateral.postprocess_minarearect_multi (:423-530): ``(clean_bin uint8 [h, w],
    detections)``, each detection ``{"label", "area", "box" int32 [4, 2] clockwise, "center", "d1", "d2", "d_mean"}`` (+ diagnostics).
    ``img_bgr`` is accepted and ignored, as in the reference."""
    if isinstance(prob01, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(prob01, dtype=np.float32)).to(torch.device(device if device is not None else "cuda"))
    else:
        t = prob01
    clean, dets = postprocess_quadrilateral_batch(t.reshape(1, *t.shape[-2:]), bin_thresh, min_area_frac, morph_kernel, open_iter, close_iter,
                                                  fit_outset_px)
    return clean[0].cpu().numpy(), dets[0]
