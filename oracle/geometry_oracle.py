"""TEST INFRASTRUCTURE — CPU restatement (numpy / scipy) of the reference's geometry post-processing, the product's actual
output (indentation diagonals in pixels -> hardness):

    ui_infer_rectangle.py:291-381   postprocess_minarearect_multi   (threshold 0.50, minAreaRect)
    ui_infer_quadrilateral.py:446-530 shares steps 1-3 (threshold 0.45) and replaces minAreaRect by a 4-vertex fit

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(vickers-hardness-unet_amd/geometry.py -> csrc/geometry.hip) never does.

**Parity unpinned for the OpenCV calls.**  The reference delegates every step to cv2 (threshold by numpy, then
cv2.getStructuringElement / morphologyEx / connectedComponentsWithStats / findContours / minAreaRect / boxPoints); cv2 is not
importable here and cannot be fetched, and the reference holds no mask, label image or detection list to compare with (its
runs/imgs/*.jpg are screenshots).  What this file restates, and what pins it:
  * binarisation `(prob01 >= bin_thresh)` (ui_infer_rectangle.py:325): numpy semantics, float32 compare — exact.
  * MORPH_ELLIPSE structuring element: OpenCV's published formula (morph.dispatch.cpp getStructuringElement: row i spans
    c +- saturate_cast<int>(c * sqrt((r*r - dy*dy) / r*r))) -> 3x3 is the CROSS, 5x5 is the square minus its four corners.
    Pinned by those two hand-written matrices (tests/test_geometry_cpu.py).
  * morphologyEx OPEN / CLOSE with `iterations` (erode^n dilate^n / dilate^n erode^n) and OpenCV's default constant border
    (morphologyDefaultBorderValue: pixels outside the image never win a min or a max).  Cross-checked against
    scipy.ndimage.binary_opening / binary_closing, an independent implementation.
  * 8-connected components with pixel areas; label ids 1..N in raster order of each component's first pixel (what
    scipy.ndimage.label produces; cv2's ids follow its block-based scan and agree except in contrived block-order cases —
    the reference only uses the id as a tag and sorts by area).
  * minimum-area enclosing rectangle of a component = of its convex hull (rotating calipers: one side collinear with a hull
    edge).  cv2.minAreaRect's float32 round-off, tie-breaking between equal-area candidates and its (w, h, angle) convention
    cannot be reproduced without the library; the corner SET agrees up to that round-off, the reference itself notes the corner
    order is not fixed (ui_infer_rectangle.py:308), and box coordinates are truncated to int32 (ui_infer_rectangle.py:349) so a
    corner may differ by one pixel from cv2's.  Pinned by hand-worked cases: axis-aligned squares / rectangles and 45-degree
    diamonds whose enclosing rectangle, area and diagonals are known in closed form.
  * diagonals: longest of the six pair distances of the int32 corners, the other diagonal from the two remaining corners
    (ui_infer_rectangle.py:352-364), float64 as numpy computes them — exact.
The arithmetic of the rectangle search is written in float32 in a fixed order so that the HIP kernel (csrc/geometry.hip,
floating-point contraction off) can match it bit for bit; the hull here is built by a different algorithm (Andrew's monotone
chain over ALL pixels) than on the device (per-row extremes), so hull agreement is a real check."""
from __future__ import annotations

import numpy as np
from scipy import ndimage

BIN_THRESH_RECT = 0.50      # ui_infer_rectangle.py:45
BIN_THRESH_QUAD = 0.45      # ui_infer_quadrilateral.py:46
MIN_AREA_FRAC = 0.0008      # :46 / :47
MORPH_KERNEL = 3
OPEN_ITER = 1
CLOSE_ITER = 1


def ellipse_kernel(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) [OpenCV formula restated]: uint8 [k][k]."""
    r, c = k // 2, k // 2
    out = np.zeros((k, k), dtype=np.uint8)
    if k == 1:
        out[0, 0] = 1
        return out
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))      # saturate_cast<int> rounds to nearest
            j1, j2 = max(c - dx, 0), min(c + dx + 1, k)
            out[i, j1:j2] = 1
    return out


def _shifted(mask: np.ndarray, dy: int, dx: int, fill: int) -> np.ndarray:
    h, w = mask.shape
    out = np.full_like(mask, fill)
    ys, yd = (slice(dy, h), slice(0, h - dy)) if dy >= 0 else (slice(0, h + dy), slice(-dy, h))
    xs, xd = (slice(dx, w), slice(0, w - dx)) if dx >= 0 else (slice(0, w + dx), slice(-dx, w))
    out[yd, xd] = mask[ys, xs]
    return out


def erode(mask: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """cv2.erode with the default border: outside pixels count as 255 (never the minimum)."""
    r = kern.shape[0] // 2
    out = np.full_like(mask, 255)
    for i in range(kern.shape[0]):
        for j in range(kern.shape[1]):
            if kern[i, j]:
                out = np.minimum(out, _shifted(mask, i - r, j - r, 255))
    return out


def dilate(mask: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """cv2.dilate with the default border: outside pixels count as 0 (never the maximum).  The kernels here are symmetric,
    so the reflection cv2 applies to the structuring element does not matter."""
    r = kern.shape[0] // 2
    out = np.zeros_like(mask)
    for i in range(kern.shape[0]):
        for j in range(kern.shape[1]):
            if kern[i, j]:
                out = np.maximum(out, _shifted(mask, i - r, j - r, 0))
    return out


def binarize(prob01: np.ndarray, bin_thresh: float) -> np.ndarray:
    """ui_infer_rectangle.py:325: (prob01 >= bin_thresh).astype(uint8) * 255; the float32 map is compared in float32."""
    return (prob01.astype(np.float32) >= np.float32(bin_thresh)).astype(np.uint8) * np.uint8(255)


def open_close(mask: np.ndarray, morph_kernel: int = MORPH_KERNEL, open_iter: int = OPEN_ITER, close_iter: int = CLOSE_ITER) -> np.ndarray:
    """ui_infer_rectangle.py:327-332: morphologyEx(OPEN, iterations=n) = n erosions then n dilations; CLOSE the other way round."""
    k = ellipse_kernel(morph_kernel)
    m = mask
    for _ in range(max(0, open_iter)):
        m = erode(m, k)
    for _ in range(max(0, open_iter)):
        m = dilate(m, k)
    for _ in range(max(0, close_iter)):
        m = dilate(m, k)
    for _ in range(max(0, close_iter)):
        m = erode(m, k)
    return m


def label8(mask: np.ndarray):
    """8-connected components: (labels int32 [h][w] with ids 1..N in raster order of the first pixel, areas int64 [N + 1])."""
    labels, n = ndimage.label(mask > 0, structure=np.ones((3, 3), dtype=np.int32))
    areas = np.bincount(labels.ravel(), minlength=n + 1)
    return labels.astype(np.int32), areas


def convex_hull(points_xy: np.ndarray) -> np.ndarray:
    """Andrew's monotone chain over integer points [n][2] (x, y); strictly convex vertices.  Returned in the canonical order the
    device uses: starting at the top-most (then left-most) vertex, down the LEFT side first (in image coordinates that is
    counter-clockwise on screen)."""
    pts = np.unique(points_xy.astype(np.int64), axis=0)          # sorted by x, then y
    if len(pts) <= 2:
        hull = pts
    else:
        def cross(o, a, b):
            return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
        lower, upper = [], []
        for p in pts:
            while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
                lower.pop()
            lower.append(tuple(p))
        for p in pts[::-1]:
            while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
                upper.pop()
            upper.append(tuple(p))
        hull = np.array(lower[:-1] + upper[:-1], dtype=np.int64)
    if len(hull) <= 1:
        return hull
    # canonical start and direction
    start = min(range(len(hull)), key=lambda i: (hull[i][1], hull[i][0]))
    hull = np.roll(hull, -start, axis=0)
    if len(hull) >= 3:
        # signed area (shoelace) in image coordinates: going top -> down the left side -> bottom -> up the right side is the
        # orientation with NEGATIVE shoelace sum when y grows downwards; flip otherwise
        x, y = hull[:, 0], hull[:, 1]
        s = int(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
        if s > 0:
            hull = np.concatenate([hull[:1], hull[1:][::-1]])
    return hull


F = np.float32


def min_area_rect(hull: np.ndarray):
    """Rotating calipers in float32 with a fixed operation order (mirrored by csrc/geometry.hip k_geom_rect):
    for every hull edge i -> unit direction u, normal v = (-uy, ux); extents of all vertices along u and v; the edge with the
    smallest (smax - smin) * (tmax - tmin) wins, first index on ties.
    Returns dict(center (cx, cy), size (along edge, across), u (ux, uy), corners float32 [4][2])."""
    m = len(hull)
    hx, hy = hull[:, 0].astype(F), hull[:, 1].astype(F)
    if m == 1:
        c = np.array([[hx[0], hy[0]]] * 4, dtype=F)
        return dict(center=(hx[0], hy[0]), size=(F(0), F(0)), u=(F(1), F(0)), corners=c)
    best = None
    edges = range(1) if m == 2 else range(m)
    for i in edges:
        j = (i + 1) % m
        dx, dy = F(hx[j] - hx[i]), F(hy[j] - hy[i])
        ln = np.sqrt(F(F(dx * dx) + F(dy * dy)), dtype=F)
        ux, uy = F(dx / ln), F(dy / ln)
        vx, vy = F(-uy), ux
        s = (hx * ux).astype(F) + (hy * uy).astype(F)
        t = (hx * vx).astype(F) + (hy * vy).astype(F)
        smin, smax, tmin, tmax = s.min(), s.max(), t.min(), t.max()
        area = F(F(smax - smin) * F(tmax - tmin))
        if best is None or area < best[0]:
            best = (area, ux, uy, vx, vy, smin, smax, tmin, tmax)
    _, ux, uy, vx, vy, smin, smax, tmin, tmax = best
    sc, tc = F(F(smin + smax) * F(0.5)), F(F(tmin + tmax) * F(0.5))
    cx = F(F(sc * ux) + F(tc * vx))
    cy = F(F(sc * uy) + F(tc * vy))
    corners = np.array([[F(F(a * ux) + F(b * vx)), F(F(a * uy) + F(b * vy))]
                        for a, b in ((smin, tmin), (smax, tmin), (smax, tmax), (smin, tmax))], dtype=F)
    return dict(center=(cx, cy), size=(F(smax - smin), F(tmax - tmin)), u=(ux, uy), corners=corners)


def diagonals(box: np.ndarray):
    """ui_infer_rectangle.py:352-364 on the int32 corners: the longest pair distance is one diagonal, the two remaining
    corners give the other."""
    pairs = []
    for a in range(4):
        for b in range(a + 1, 4):
            pairs.append((float(np.linalg.norm(box[a] - box[b])), a, b))
    pairs.sort(reverse=True, key=lambda x: x[0])
    _, i1, j1 = pairs[0]
    rest = [k for k in range(4) if k not in (i1, j1)]
    d1 = float(np.linalg.norm(box[i1] - box[j1]))
    d2 = float(np.linalg.norm(box[rest[0]] - box[rest[1]]))
    return d1, d2


def min_area(h: int, w: int, min_area_frac: float = MIN_AREA_FRAC) -> int:
    return max(200, int(min_area_frac * h * w))        # ui_infer_rectangle.py:322


def postprocess_minarearect_multi(prob01: np.ndarray, bin_thresh: float = BIN_THRESH_RECT, min_area_frac: float = MIN_AREA_FRAC,
                                  morph_kernel: int = MORPH_KERNEL, open_iter: int = OPEN_ITER, close_iter: int = CLOSE_ITER):
    """ui_infer_rectangle.py:291-381 (the image argument of the reference is unused by it).  Returns (clean uint8 [h][w] in
    {0, 255}, detections sorted by area, largest first: dict(label, area, box int32 [4][2], center, d1, d2, d_mean))."""
    h, w = prob01.shape[:2]
    amin = min_area(h, w, min_area_frac)
    mask = open_close(binarize(prob01, bin_thresh), morph_kernel, open_iter, close_iter)
    labels, areas = label8(mask)
    clean = np.zeros_like(mask)
    dets = []
    for i in range(1, len(areas)):
        area = int(areas[i])
        if area < amin:
            continue
        sel = labels == i
        clean[sel] = 255
        ys, xs = np.nonzero(sel)
        hull = convex_hull(np.stack([xs, ys], axis=1))
        rect = min_area_rect(hull)
        box = rect["corners"].astype(np.int32)            # boxPoints(...).astype(np.int32): truncation (ui_infer_rectangle.py:349)
        d1, d2 = diagonals(box)
        dets.append(dict(label=i, area=area, box=box, center=(float(rect["center"][0]), float(rect["center"][1])),
                         d1=d1, d2=d2, d_mean=0.5 * (d1 + d2), rect=rect, hull=hull))
    dets.sort(key=lambda d: d["area"], reverse=True)      # stable: equal areas keep label order (ui_infer_rectangle.py:379)
    return clean, dets
