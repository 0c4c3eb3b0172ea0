"""TEST INFRASTRUCTURE — CPU restatement (numpy, pure-Python loops) of the 4-vertex fit of the newer GUI, the product's actual
output (ui_infer_quadrilateral.py):

    :262-330   _order_quad_cw, _is_convex_quad, _poly_area, _quad_quality
    :331-420   robust_quadrilateral_from_contour   (hull, approxPolyDP eps-bisection to 4 points, fall-backs, quality sort)
    :423-530   postprocess_minarearect_multi       (threshold 0.45, open/close, components, 2-px outset dilation, per-component fit,
                                                    diagonals)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(vickers-hardness-unet_amd/geometry.py -> csrc/geometry.hip) never does.

**Parity unpinned for the OpenCV calls.**  cv2 (un-pinned third-party dependency of the reference) is not importable here and
the reference holds no contour, polygon or detection list.  Steps 1-3 are shared with oracle/geometry_oracle.py.  What this file
restates from OpenCV 4.x's published sources, and what pins it:
  * cv2.dilate with the (2 * fit_outset_px + 1)^2 MORPH_ELLIPSE element (geometry_oracle.ellipse_kernel / dilate).
  * cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) on the dilation of ONE 8-connected component (one external border):
    Suzuki-Abe border following as contours.cpp `icvFetchContour` runs it — direction codes 0..7 = E, NE, N, NW, W, SW, S, SE
    (y grows downwards), start at the raster-first pixel, first neighbour searched clockwise from W, then every next border pixel
    searched counter-clockwise from the direction after the one we came from; a point is emitted where the chain code changes.
    Pinned by hand: a filled rectangle gives its corners in the order TL, BL, BR, TR; a 45-degree diamond its four tips.
  * cv2.convexHull(points) (default clockwise=False, y up): strictly convex vertices, starting at the right-most (then bottom-most)
    point, running clockwise ON SCREEN (right -> bottom -> left -> top).  The vertex set is pinned against the monotone chain
    of geometry_oracle.convex_hull over all pixels; the start / direction only matter for exact distance ties in what follows.
  * cv2.arcLength: float32 segment lengths accumulated in float64.
  * cv2.approxPolyDP (approx.cpp `approxPolyDP_`, closed curve): three farthest-point sweeps choose the two anchors, an explicit
    stack of index slices splits at the farthest point while max_dist^2 > eps^2 * |chord|^2 (float64 on integer-valued
    coordinates: every distance is exact), and a final in-place pass drops a vertex that lies within sqrt(eps^2 / 2) of the chord of
    its neighbours when that chord is neither horizontal nor vertical and the turn is not a reversal.  Pinned by closed-form
    shapes: noisy rectangles / diamonds collapse to their four corners, a regular octagon needs the sub-sampling branch.
  * numpy pieces of the reference (float32 arctan2 ordering, float32 shoelace, the quality score's mixed float32 / float64
    arithmetic, Python's stable sort on (quality, area)) are restated operation by operation.
The device kernel (csrc/geometry.hip k_geom_quad) follows the same operation order in the same types with floating-point
contraction off, so device and oracle agree bit for bit on the int32 corners and float64 diagonals."""
from __future__ import annotations

import math

import numpy as np

from . import geometry_oracle as G

F = np.float32
FIT_OUTSET_PX = 2          # ui_infer_quadrilateral.py:433 default
MAX_ITER = 25              # :332


# ------------------------------------------------------------------------------------------------ :262-330
def order_quad_cw(pts) -> np.ndarray:
    """:266-277 — clockwise on screen (descending arctan2 about the float32 mean), rotated to start at min y, then min x."""
    p = np.asarray(pts).astype(F).reshape(-1, 2)
    c = (((p[0] + p[1]) + p[2]) + p[3]) / F(4) if len(p) == 4 else p.mean(axis=0)     # float32 row-by-row sum, as np.mean reduces axis 0
    ang = np.arctan2(p[:, 1] - c[1], p[:, 0] - c[0])
    idx = np.argsort(ang, kind="stable")
    p = p[idx[::-1]]
    k = np.lexsort((p[:, 0], p[:, 1]))[0]
    return np.roll(p, -k, axis=0)


def is_convex_quad(p) -> bool:
    """:280-295 — the four float32 cross products of consecutive edges share a sign (zeros allowed)."""
    p = np.asarray(p, dtype=F).reshape(4, 2)
    sg = []
    for i in range(4):
        a, b, c = p[i], p[(i + 1) % 4], p[(i + 2) % 4]
        v1, v2 = b - a, c - b
        sg.append(F(F(v1[0] * v2[1]) - F(v1[1] * v2[0])))
    return all(x >= 0 for x in sg) or all(x <= 0 for x in sg)


def poly_area(p) -> float:
    """:298-301 — float32 shoelace (the two 4-term dot products summed left to right), abs, halved in float64."""
    p = np.asarray(p, dtype=F).reshape(-1, 2)
    x, y = p[:, 0], p[:, 1]
    yr, xr = np.roll(y, -1), np.roll(x, -1)
    s1, s2 = F(0), F(0)
    for i in range(len(p)):
        s1 = F(s1 + F(x[i] * yr[i]))
        s2 = F(s2 + F(y[i] * xr[i]))
    return abs(float(F(s1 - s2))) * 0.5


def quad_quality(p) -> float:
    """:304-330 — (1 - angle penalty / 2)(1 - side-ratio penalty / 2) * perimeter / (perimeter + 1000); float32 lengths and
    cosines, float64 angle test and final product."""
    p = np.asarray(p, dtype=F).reshape(4, 2)

    def norm(v):
        return F(np.sqrt(F(F(v[0] * v[0]) + F(v[1] * v[1]))))

    d = [norm(p[i] - p[(i + 1) % 4]) for i in range(4)]
    peri = F(F(F(F(d[0] + d[1]) + d[2]) + d[3]) + F(1e-6))
    pen = []
    for i in range(4):
        a, b, c = p[(i - 1) % 4], p[i], p[(i + 1) % 4]
        v1, v2 = a - b, c - b
        dot = F(F(v1[0] * v2[0]) + F(v1[1] * v2[1]))
        cs = F(dot / F(F(norm(v1) * norm(v2)) + F(1e-6)))
        ang = math.degrees(math.acos(float(min(max(cs, F(-1)), F(1)))))
        pen.append(0.0 if 15.0 <= ang <= 165.0 else 1.0)
    ang_pen = (pen[0] + pen[1] + pen[2] + pen[3]) / 4.0
    ratio = F(F(max(d) + F(1e-6)) / F(min(d) + F(1e-6)))
    ed_pen = min(1.0, float(abs(F(ratio - F(1.0)))))
    shape = float(F(F(1.0) - F(F(0.5) * F(ed_pen))))
    size = float(F(peri / F(peri + F(1000.0))))
    return (1.0 - 0.5 * ang_pen) * shape * size


# ------------------------------------------------------------------------------------------------ cv2 pieces
def trace_external_contour(mask: np.ndarray) -> np.ndarray:
    """cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) for a mask holding ONE 8-connected component: int32 [n][2]
    (x, y).  [OpenCV contours.cpp icvFetchContour restated]"""
    h, w = mask.shape
    ys, xs = np.nonzero(mask)
    if len(ys) == 0:
        return np.zeros((0, 2), dtype=np.int32)
    y0 = int(ys.min())
    x0 = int(xs[ys == y0].min())
    DX = (1, 1, 0, -1, -1, -1, 0, 1)
    DY = (0, -1, -1, -1, 0, 1, 1, 1)

    def on(x, y):
        return 0 <= x < w and 0 <= y < h and mask[y, x] != 0

    s, s_end = 4, 4
    while True:                                   # first neighbour, clockwise from W
        s = (s - 1) & 7
        if on(x0 + DX[s], y0 + DY[s]) or s == s_end:
            break
    if s == s_end and not on(x0 + DX[s], y0 + DY[s]):
        return np.array([[x0, y0]], dtype=np.int32)
    x1, y1 = x0 + DX[s], y0 + DY[s]
    out = []
    x3, y3 = x0, y0
    prev_s = s ^ 4
    px, py = x0, y0
    while True:
        while True:                               # next border pixel, counter-clockwise from the direction after the incoming one
            s = (s + 1) & 7
            x4, y4 = x3 + DX[s], y3 + DY[s]
            if on(x4, y4):
                break
        if s != prev_s:
            out.append((px, py))
            prev_s = s
        px += DX[s]
        py += DY[s]
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return np.array(out, dtype=np.int32)


def convex_hull_cv(points_xy: np.ndarray) -> np.ndarray:
    """cv2.convexHull(points) with the default orientation: strictly convex vertices from the right-most (then bottom-most)
    point, clockwise on screen."""
    hull = G.convex_hull(points_xy)              # from the top-most vertex, counter-clockwise on screen
    if len(hull) <= 2:
        hull = hull[np.lexsort((hull[:, 1], hull[:, 0]))][::-1] if len(hull) == 2 else hull
        return hull.astype(np.int32)
    hull = hull[::-1]                            # clockwise on screen
    start = max(range(len(hull)), key=lambda i: (hull[i][0], hull[i][1]))
    return np.roll(hull, -start, axis=0).astype(np.int32)


def arc_length_closed(poly: np.ndarray) -> float:
    """cv2.arcLength(poly, True): float32 segment lengths, float64 sum, starting with the closing segment."""
    p = np.asarray(poly, dtype=F).reshape(-1, 2)
    if len(p) <= 1:
        return 0.0
    per = 0.0
    prev = p[-1]
    for q in p:
        dx, dy = F(q[0] - prev[0]), F(q[1] - prev[1])
        per += float(F(np.sqrt(F(F(dx * dx) + F(dy * dy)))))
        prev = q
    return per


def approx_poly_dp_closed(poly: np.ndarray, epsilon: float) -> np.ndarray:
    """cv2.approxPolyDP(poly, epsilon, closed=True) on integer-valued points.  [OpenCV approx.cpp approxPolyDP_ restated]"""
    src = [(float(x), float(y)) for x, y in np.asarray(poly).reshape(-1, 2)]
    count = len(src)
    if count == 0:
        return np.zeros((0, 2), dtype=F)
    eps = float(epsilon) * float(epsilon)
    dst = []
    stack = []
    # 1. two (approximately) farthest points: three sweeps, each restarting from the farthest point of the previous one
    pos, right_start, le_eps = 0, 0, False
    start_pt = src[0]
    for _ in range(3):
        pos = (pos + right_start) % count
        start_pt = src[pos]
        pos = (pos + 1) % count
        max_dist = 0.0
        for j in range(1, count):
            pt = src[pos]
            pos = (pos + 1) % count
            dx, dy = pt[0] - start_pt[0], pt[1] - start_pt[1]
            dist = dx * dx + dy * dy
            if dist > max_dist:
                max_dist = dist
                right_start = j
        le_eps = max_dist <= eps
    # 2. initial slices
    if not le_eps:
        a = pos % count
        b = (right_start + a) % count
        stack.append((b, a))          # right slice (pushed first, popped last)
        stack.append((a, b))
    else:
        dst.append(start_pt)
    # 3. split at the farthest point while it is farther than eps from the chord
    while stack:
        s0, s1 = stack.pop()
        end_pt = src[s1]
        pos = s0
        start_pt = src[pos]
        pos = (pos + 1) % count
        split = 0
        if pos != s1:
            dx, dy = end_pt[0] - start_pt[0], end_pt[1] - start_pt[1]
            max_dist = 0.0
            while pos != s1:
                pt = src[pos]
                pos = (pos + 1) % count
                dist = abs((pt[1] - start_pt[1]) * dx - (pt[0] - start_pt[0]) * dy)
                if dist > max_dist:
                    max_dist = dist
                    split = (pos + count - 1) % count
            le = max_dist * max_dist <= eps * (dx * dx + dy * dy)
        else:
            le = True
        if le:
            dst.append(start_pt)
        else:
            stack.append((split, s1))
            stack.append((s0, split))
    # 4. in-place clean-up of nearly straight runs (reads can see already rewritten entries near the wrap-around, as in OpenCV)
    count = new_count = len(dst)
    if count == 0:
        return np.zeros((0, 2), dtype=F)
    pos = count - 1
    start_pt = dst[pos]
    pos = (pos + 1) % count
    wpos = pos
    pt = dst[pos]
    pos = (pos + 1) % count
    i = 0
    while i < count and new_count > 2:
        end_pt = dst[pos]
        pos = (pos + 1) % count
        dx, dy = end_pt[0] - start_pt[0], end_pt[1] - start_pt[1]
        dist = abs((pt[0] - start_pt[0]) * dy - (pt[1] - start_pt[1]) * dx)
        inner = (pt[0] - start_pt[0]) * (end_pt[0] - pt[0]) + (pt[1] - start_pt[1]) * (end_pt[1] - pt[1])
        if dist * dist <= 0.5 * eps * (dx * dx + dy * dy) and dx != 0 and dy != 0 and inner >= 0:
            new_count -= 1
            dst[wpos] = start_pt = end_pt
            wpos = (wpos + 1) % count
            pt = dst[pos]
            pos = (pos + 1) % count
            i += 2
            continue
        dst[wpos] = start_pt = pt
        wpos = (wpos + 1) % count
        pt = end_pt
        i += 1
    return np.array(dst[:new_count], dtype=F).reshape(-1, 2)


# ------------------------------------------------------------------------------------------------ :331-420
def robust_quadrilateral_from_contour(cnt: np.ndarray, want_convex: bool = True, max_iter: int = MAX_ITER, trace: dict | None = None):
    """:336-420.  `trace` (optional dict) receives which branch produced the candidates."""
    pts = np.asarray(cnt).reshape(-1, 2).astype(F)
    if pts.shape[0] < 4:
        return None
    hull = convex_hull_cv(pts.astype(np.int32)).astype(F)

    def ok(cand):
        return poly_area(cand) > 10 and (not want_convex or is_convex_quad(cand))

    def try_poly_dp(poly):
        peri = arc_length_closed(poly)
        lo, hi = 0.001 * peri, 0.08 * peri
        for _ in range(max_iter):
            mid = 0.5 * (lo + hi)
            appr = approx_poly_dp_closed(poly, mid)
            n = len(appr)
            if n == 4:
                cand = order_quad_cw(appr)
                if ok(cand):
                    return cand
                lo = mid
            elif n > 4:
                lo = mid
            else:
                hi = mid
            if abs(hi - lo) < 1e-6:
                break
        return None

    cands, branch = [], "bisection"
    for poly in (pts, hull):
        got = try_poly_dp(poly)
        if got is not None:
            cands.append(got)
    if not cands:
        branch = "subsample"
        for poly in (pts, hull):
            appr = approx_poly_dp_closed(poly, 0.01 * arc_length_closed(poly))
            k = len(appr)
            if k > 4:
                for s in range(min(12, k)):
                    cand = order_quad_cw(appr[np.arange(s, s + 4) % k])
                    if ok(cand):
                        cands.append(cand)
    if not cands:
        branch = "extremes"
        xs, ys = hull[:, 0], hull[:, 1]
        raw = np.array([hull[int(np.argmin(ys))], hull[int(np.argmax(xs))], hull[int(np.argmax(ys))], hull[int(np.argmin(xs))]], dtype=F)
        cand = order_quad_cw(raw)
        if poly_area(cand) > 10:
            cands.append(cand)
    if trace is not None:
        trace["branch"] = branch if cands else "none"
        trace["n_candidates"] = len(cands)
    if not cands:
        return None
    best, best_key = None, None
    for q in cands:                                  # sort(key, reverse=True)[0] of a stable sort: the FIRST maximal key wins
        key = (quad_quality(q), poly_area(q))
        if best is None or key > best_key:
            best, best_key = q, key
    return best


# ------------------------------------------------------------------------------------------------ :423-530
def quad_diagonals(box: np.ndarray):
    return G.diagonals(box)                           # same rule as the rectangle GUI (:503-512)


def postprocess_quadrilateral_multi(prob01: np.ndarray, bin_thresh: float = G.BIN_THRESH_QUAD, min_area_frac: float = G.MIN_AREA_FRAC,
                                    morph_kernel: int = G.MORPH_KERNEL, open_iter: int = G.OPEN_ITER, close_iter: int = G.CLOSE_ITER,
                                    fit_outset_px: int = FIT_OUTSET_PX):
    """ui_infer_quadrilateral.py:423-530 `postprocess_minarearect_multi` (its image argument is unused).  Returns (clean uint8
    [h][w] in {0, 255}, detections sorted by area: dict(label, area, box int32 [4][2] clockwise from the top-most corner, center,
    d1, d2, d_mean) + diagnostic keys contour / hull / branch)."""
    h, w = prob01.shape[:2]
    amin = G.min_area(h, w, min_area_frac)
    mask = G.open_close(G.binarize(prob01, bin_thresh), morph_kernel, open_iter, close_iter)
    labels, areas = G.label8(mask)
    clean = np.zeros_like(mask)
    k_fit = G.ellipse_kernel(max(3, fit_outset_px * 2 + 1)) if fit_outset_px > 0 else None
    dets = []
    for i in range(1, len(areas)):
        area = int(areas[i])
        if area < amin:
            continue
        sel = labels == i
        clean[sel] = 255
        mask_fit = sel.astype(np.uint8) * np.uint8(255)
        if k_fit is not None:
            mask_fit = G.dilate(mask_fit, k_fit)
        cnt = trace_external_contour(mask_fit)
        tr = {}
        quad = robust_quadrilateral_from_contour(cnt, True, MAX_ITER, tr)
        if quad is None:
            continue
        box = order_quad_cw(quad).astype(np.int32)
        cx, cy = float(np.mean(box[:, 0])), float(np.mean(box[:, 1]))
        d1, d2 = quad_diagonals(box)
        dets.append(dict(label=i, area=area, box=box, center=(cx, cy), d1=d1, d2=d2, d_mean=0.5 * (d1 + d2), contour=cnt,
                         hull=convex_hull_cv(cnt), branch=tr.get("branch"), n_candidates=tr.get("n_candidates")))
    dets.sort(key=lambda d: d["area"], reverse=True)
    return clean, dets
