"""oracle/quad_oracle.py (the 4-vertex fit of ui_infer_quadrilateral.py:262-530) pinned on shapes with closed-form answers.
cv2 is absent (parity with OpenCV itself: unpinned, see the oracle's header); what CAN be pinned is pinned here:
border-following order on a rectangle and a diamond, Douglas-Peucker on polygons whose answer is known, the ordering /
convexity / area / quality helpers against hand-worked numbers, the fall-back branch on a pentagon, and plausibility on the
reference's own label masks (tests/golden/real_masks.npz: the fitted diagonals agree with the minimum-area rectangle's within 3 %)."""
import math

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import geometry_oracle as G
from oracle import quad_oracle as Q


def _real_masks():
    z = np.load(GOLDEN / "real_masks.npz")
    for i, nm in enumerate(z["names"]):
        h, w = (int(v) for v in z[f"shape_{i}"])
        yield str(nm), np.unpackbits(z[f"bits_{i}"], axis=1)[:, :w].astype(np.float32).reshape(h, w)


def test_border_following_order_and_simple_approximation():
    m = np.zeros((20, 30), np.uint8)
    m[5:12, 8:20] = 255
    # cv2.findContours on a filled rectangle: TL, BL, BR, TR (outer borders run counter-clockwise on screen)
    assert Q.trace_external_contour(m).tolist() == [[8, 5], [8, 11], [19, 11], [19, 5]]
    d = np.zeros((41, 41), np.uint8)
    yy, xx = np.mgrid[0:41, 0:41]
    d[np.abs(xx - 20) + np.abs(yy - 20) <= 10] = 255
    c = Q.trace_external_contour(d)
    assert c.tolist() == [[20, 10], [10, 20], [20, 30], [30, 20]]          # pure diagonal runs collapse to the four tips
    assert Q.convex_hull_cv(c).tolist() == [[30, 20], [20, 30], [10, 20], [20, 10]]      # from the right-most vertex, clockwise on screen
    one = np.zeros((5, 5), np.uint8)
    one[2, 3] = 255
    assert Q.trace_external_contour(one).tolist() == [[3, 2]]
    # a one-pixel-wide line is walked out and back
    ln = np.zeros((5, 9), np.uint8)
    ln[2, 1:8] = 255
    assert Q.trace_external_contour(ln).tolist() == [[1, 2], [7, 2]]
    # every contour point is a border pixel of the mask and the hull of the contour is the hull of the mask
    rng = np.random.default_rng(3)
    blob = G.open_close((rng.random((64, 64)) > 0.35).astype(np.uint8) * 255)
    lab, areas = G.label8(blob)
    k = int(np.argmax(areas[1:])) + 1
    comp = (lab == k).astype(np.uint8) * 255
    cnt = Q.trace_external_contour(comp)
    assert all(comp[y, x] for x, y in cnt)
    ys, xs = np.nonzero(comp)
    assert sorted(map(tuple, Q.convex_hull_cv(cnt).tolist())) == sorted(map(tuple, G.convex_hull(np.stack([xs, ys], 1)).tolist()))


def test_arc_length_and_douglas_peucker_closed_forms():
    sq = np.array([[0, 0], [0, 10], [10, 10], [10, 0]], np.int32)
    assert Q.arc_length_closed(sq) == 40.0
    assert Q.arc_length_closed(np.array([[0, 0], [3, 4]], np.int32)) == 10.0
    # a square with a 1-pixel dent in every side: epsilon 0.5 keeps the dents, 1.5 removes them, 8 collapses further
    dent = np.array([[0, 0], [0, 5], [1, 5], [0, 6], [0, 10], [5, 10], [5, 9], [6, 10], [10, 10], [10, 5], [9, 5], [10, 4], [10, 0], [5, 0], [5, 1], [4, 0]], np.int32)
    assert len(Q.approx_poly_dp_closed(dent, 0.5)) == 16
    four = Q.approx_poly_dp_closed(dent, 1.5)
    assert sorted(map(tuple, four.tolist())) == [(0.0, 0.0), (0.0, 10.0), (10.0, 0.0), (10.0, 10.0)]
    assert len(Q.approx_poly_dp_closed(dent, 8.0)) <= 3
    # staircase of a 45-degree edge: within epsilon 1 of the chord -> the triangle's three corners
    tri = np.array([[0, 0], [0, 20], [20, 20]] + [[20 - i - (1 if k else 0), 20 - i - 1] for i in range(19) for k in (0, 1)][:-1], np.int32)
    out = Q.approx_poly_dp_closed(tri, 1.0)
    assert sorted(map(tuple, out.tolist())) == [(0.0, 0.0), (0.0, 20.0), (20.0, 20.0)]
    # degenerate inputs
    assert len(Q.approx_poly_dp_closed(np.zeros((0, 2), np.int32), 1.0)) == 0
    assert Q.approx_poly_dp_closed(np.array([[3, 4]], np.int32), 1.0).tolist() == [[3.0, 4.0]]


def test_ordering_convexity_area_quality():
    q = Q.order_quad_cw(np.array([[10, 0], [0, 10], [10, 20], [20, 10]], np.float32))
    assert q.tolist() == [[10.0, 0.0], [0.0, 10.0], [10.0, 20.0], [20.0, 10.0]]            # starts at min y; descending arctan2 (y down)
    q2 = Q.order_quad_cw(np.array([[20, 10], [10, 20], [0, 10], [10, 0]], np.float32))
    assert q2.tolist() == q.tolist()                                                         # any input order, one output order
    sq = Q.order_quad_cw(np.array([[0, 0], [10, 0], [10, 10], [0, 10]], np.float32))
    assert sq[0].tolist() == [0.0, 0.0]                                                      # y tie -> min x
    assert Q.is_convex_quad(q) and Q.poly_area(q) == 200.0
    dart = np.array([[0, 0], [10, 4], [20, 0], [10, 20]], np.float32)                        # reflex vertex at (10, 4)
    assert not Q.is_convex_quad(dart)
    # square of side s: no penalties -> quality = peri / (peri + 1000) in float32
    s = 100.0
    quality = Q.quad_quality(np.array([[0, 0], [0, s], [s, s], [s, 0]], np.float32))
    assert quality == pytest.approx(400.0 / 1400.0, rel=1e-6)
    # 2:1 rectangle: side-ratio penalty min(1, |2 - 1|) = 1 -> factor 0.5
    r = Q.quad_quality(np.array([[0, 0], [0, 50], [100, 50], [100, 0]], np.float32))
    assert r == pytest.approx(0.5 * 300.0 / 1300.0, rel=1e-6)
    # a rhombus with 10 / 170-degree corners: all four angles outside [15, 165] -> angle factor 0.5 (sides equal: no ratio penalty);
    # with 20 / 160-degree corners nothing is penalised
    for half_angle, factor in ((5.0, 0.5), (10.0, 1.0)):
        a = math.radians(half_angle)
        rh = np.array([[0, 0], [100 * math.cos(a), 100 * math.sin(a)], [200 * math.cos(a), 0], [100 * math.cos(a), -100 * math.sin(a)]], np.float32)
        assert Q.quad_quality(rh) == pytest.approx(factor * 400.0 / 1400.0, rel=1e-4)


def _diamond_map(S, cx, cy, half, ang):
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    u = (xx - cx) * math.cos(ang) + (yy - cy) * math.sin(ang)
    v = -(xx - cx) * math.sin(ang) + (yy - cy) * math.cos(ang)
    return ((np.abs(u) + np.abs(v)) <= half).astype(np.float32)


@pytest.mark.parametrize("ang", [0.0, 0.3, math.pi / 4, 1.0, math.pi / 6])
def test_diamonds_recovered_with_the_outset(ang):
    """An indentation-shaped square of half-diagonal 60: the fit sees it dilated by the 5x5 ellipse (2 px outwards), so both
    diagonals come out at 120 + 2 * 2 within the staircase / int-truncation noise, and the vertices sit on the true corners + 2 px."""
    prob = _diamond_map(256, 120.3, 131.7, 60.0, ang)
    clean, dets = Q.postprocess_quadrilateral_multi(prob)
    assert len(dets) == 1 and dets[0]["branch"] == "bisection" and dets[0]["n_candidates"] == 2
    d = dets[0]
    assert 120.0 <= d["d2"] <= d["d1"] <= 125.5
    assert d["center"] == pytest.approx((120.3, 131.7), abs=1.5)
    box = d["box"].astype(np.float64)
    ideal = np.array([[120.3 + 62 * math.cos(ang + k * math.pi / 2), 131.7 + 62 * math.sin(ang + k * math.pi / 2)] for k in range(4)])
    for p in box:
        assert np.min(np.linalg.norm(ideal - p, axis=1)) <= 3.0
    assert box[0, 1] == box[:, 1].min()                                 # starts at the top-most corner
    assert int(clean.sum() // 255) == d["area"]
    # without the outset the same square measures ~120
    _, d0 = Q.postprocess_quadrilateral_multi(prob, fit_outset_px=0)
    assert 116.0 <= d0[0]["d2"] <= d0[0]["d1"] <= 121.5


def test_pentagon_falls_to_subsampling_and_ranking_is_stable():
    S = 160
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    pts = [(80 + 35 * math.cos(0.2 + 2 * math.pi * k / 5), 80 + 35 * math.sin(0.2 + 2 * math.pi * k / 5)) for k in range(5)]
    m = np.ones((S, S), bool)
    for i in range(5):
        a, b = pts[i], pts[(i + 1) % 5]
        m &= ((b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])) >= 0
    clean, dets = Q.postprocess_quadrilateral_multi(m.astype(np.float32))
    assert len(dets) == 1 and dets[0]["branch"] == "subsample" and dets[0]["n_candidates"] >= 5
    box = dets[0]["box"]
    assert Q.is_convex_quad(box.astype(np.float32)) and Q.poly_area(box) > 1500
    # the chosen candidate has the maximal (quality, area) key among four consecutive vertices of the 1 % polygon
    cnt = dets[0]["contour"]
    best = None
    for poly in (cnt, Q.convex_hull_cv(cnt)):
        appr = Q.approx_poly_dp_closed(poly, 0.01 * Q.arc_length_closed(poly))
        for s in range(min(12, len(appr))):
            c = Q.order_quad_cw(appr[np.arange(s, s + 4) % len(appr)])
            if Q.poly_area(c) > 10 and Q.is_convex_quad(c):
                key = (Q.quad_quality(c), Q.poly_area(c))
                best = key if best is None or key > best else best
    got = Q.order_quad_cw(box.astype(np.float32))
    assert (Q.quad_quality(got), Q.poly_area(got)) == best


def test_extreme_point_fallback():
    """A flat triangle given as four contour points: no epsilon yields four valid vertices (a 4-point result repeats a vertex or is
    not better than collinear), the 1 % polygon has fewer than five vertices, so the hull's extreme points are taken:
    [top-most, right-most, bottom-most, left-most] = a degenerate quadrilateral of area 50 > 10, which is accepted."""
    cnt = np.array([[0, 0], [50, 0], [100, 0], [50, 1]], np.int32)
    tr = {}
    q = Q.robust_quadrilateral_from_contour(cnt, trace=tr)
    assert tr == {"branch": "extremes", "n_candidates": 1}
    assert Q.poly_area(q) == 50.0 and sorted(map(tuple, q.tolist())) == [(0.0, 0.0), (50.0, 1.0), (100.0, 0.0), (100.0, 0.0)]
    assert Q.robust_quadrilateral_from_contour(cnt[:3]) is None          # fewer than four points: None (:346-347)
    # all four on one line: even the extreme points enclose nothing -> None, the component is dropped
    tr = {}
    assert Q.robust_quadrilateral_from_contour(np.array([[0, 0], [30, 0], [60, 0], [90, 0]], np.int32), trace=tr) is None and tr["branch"] == "none"


def test_real_masks_quadrilateral_vs_rectangle():
    n = 0
    for name, mask in _real_masks():
        clean_q, dq = Q.postprocess_quadrilateral_multi(mask)
        clean_r, dr = G.postprocess_minarearect_multi(mask, bin_thresh=0.45)
        assert np.array_equal(clean_q, clean_r)
        assert [d["label"] for d in dq] == [d["label"] for d in dr] and len(dq) >= 1, name
        for a, b in zip(dq, dr):
            n += 1
            assert a["branch"] == "bisection"
            assert a["d_mean"] == pytest.approx(b["d_mean"], rel=3e-2), (name, a["d_mean"], b["d_mean"])
            assert Q.is_convex_quad(a["box"].astype(np.float32))
            # the quadrilateral hugs the (2 px dilated) indentation: its area is at most the enclosing rectangle's + the dilation ring
            assert Q.poly_area(a["box"]) <= 1.06 * Q.poly_area(b["box"]) + 4 * b["d_mean"]
    assert n >= 9
