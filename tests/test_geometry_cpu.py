"""CPU tests of the geometry oracle (oracle/geometry_oracle.py): cv2 is absent, so the restatement of
ui_infer_rectangle.py:291-381 is pinned to what can be known without the library — OpenCV's published structuring elements,
an independent morphology implementation (scipy), and shapes whose minimum-area rectangle and diagonals have closed forms."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import geometry_oracle as G


def test_ellipse_kernels_are_opencv_s():
    assert G.ellipse_kernel(3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]                       # 3x3 MORPH_ELLIPSE is the cross
    assert G.ellipse_kernel(5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    k7 = G.ellipse_kernel(7)
    assert k7.shape == (7, 7) and k7[3].all() and k7[:, 3].all() and k7[0].sum() == 1 and (k7 == k7[::-1]).all() and (k7 == k7.T[::-1].T).all()


@pytest.mark.parametrize("k,oi,ci", [(3, 1, 1), (5, 1, 1), (3, 2, 1), (3, 0, 1), (7, 1, 2)])
def test_open_close_matches_scipy(k, oi, ci):
    rng = np.random.default_rng(k * 10 + oi)
    m = (ndimage.gaussian_filter(rng.normal(size=(90, 130)), 2.0) > 0.02).astype(np.uint8) * 255
    m[0:6, 0:9] = 255            # touches the border: cv2's border never erodes
    m[85:, 100:] = 255
    se = G.ellipse_kernel(k).astype(bool)
    ref = m > 0
    for _ in range(oi):
        ref = ndimage.binary_erosion(ref, structure=se, border_value=1)
    for _ in range(oi):
        ref = ndimage.binary_dilation(ref, structure=se, border_value=0)
    for _ in range(ci):
        ref = ndimage.binary_dilation(ref, structure=se, border_value=0)
    for _ in range(ci):
        ref = ndimage.binary_erosion(ref, structure=se, border_value=1)
    got = G.open_close(m, k, oi, ci)
    assert set(np.unique(got)) <= {0, 255}
    assert np.array_equal(got > 0, ref)


def test_binarize_is_float32_greater_equal():
    p = np.array([[0.5, np.nextafter(np.float32(0.5), np.float32(0)), 0.45, 0.449999]], dtype=np.float32)
    assert G.binarize(p, 0.5).tolist() == [[255, 0, 0, 0]]
    assert G.binarize(p, 0.45).tolist() == [[255, 255, 255, 0]]      # float32(0.45) >= float32(0.45): the threshold is cast, not the map


def test_label8_raster_order_and_diagonal_contact():
    m = np.zeros((6, 8), dtype=np.uint8)
    m[0, 5] = m[1, 4] = m[2, 3] = 255          # one diagonal chain: a single component under 8-connectivity
    m[0, 0] = 255                              # first pixel in raster order -> label 1
    m[4, 6:8] = 255
    lab, areas = G.label8(m)
    assert areas.tolist()[1:] == [1, 3, 2]
    assert lab[0, 0] == 1 and lab[0, 5] == lab[1, 4] == lab[2, 3] == 2 and lab[4, 7] == 3


def _prob_from_mask(m):
    return np.where(m, 0.9, 0.1).astype(np.float32)


def test_axis_aligned_rectangle_closed_form():
    m = np.zeros((80, 100), dtype=bool)
    m[5:35, 10:30] = True                      # x 10..29, y 5..34: 20 x 30 pixels
    clean, dets = G.postprocess_minarearect_multi(_prob_from_mask(m), min_area_frac=0.0)
    want = m.copy()
    for y, x in ((5, 10), (5, 29), (34, 10), (34, 29)):       # opening with the cross rounds the four corner pixels off
        want[y, x] = False
    assert np.array_equal(clean > 0, want) and len(dets) == 1
    d = dets[0]
    assert d["label"] == 1 and d["area"] == 596
    assert sorted(map(tuple, d["box"].tolist())) == [(10, 5), (10, 34), (29, 5), (29, 34)]
    assert d["center"] == pytest.approx((19.5, 19.5))
    assert d["d1"] == d["d2"] == pytest.approx(np.hypot(19, 29))
    assert len(d["hull"]) == 8                                 # the rounded corners: two hull vertices each


@pytest.mark.parametrize("R", [12, 25, 40])
def test_diamond_closed_form(R):
    """|x - cx| + |y - cy| <= R: open/close with the cross leaves an L1 ball unchanged; its hull is the four tips, the minimum
    rectangle is the diamond itself, both diagonals are 2R (truncation of the float32 corners may cost one pixel)."""
    yy, xx = np.mgrid[0:128, 0:128]
    m = (np.abs(xx - 64) + np.abs(yy - 60)) <= R
    clean, dets = G.postprocess_minarearect_multi(_prob_from_mask(m), min_area_frac=0.0)
    assert np.array_equal(clean > 0, m) and len(dets) == 1
    d = dets[0]
    assert d["area"] == 2 * R * R + 2 * R + 1
    assert [tuple(p) for p in d["hull"].tolist()] == [(64, 60 - R), (64 - R, 60), (64, 60 + R), (64 + R, 60)]     # canonical order
    tips = np.array([(64, 60 - R), (64 - R, 60), (64, 60 + R), (64 + R, 60)])
    for c in d["box"]:
        assert np.abs(tips - c).sum(axis=1).min() <= 1
    assert abs(d["d1"] - 2 * R) <= 1.5 and abs(d["d2"] - 2 * R) <= 1.5
    assert d["center"] == pytest.approx((64.0, 60.0), abs=1e-3)
    assert d["rect"]["size"][0] == pytest.approx(R * np.sqrt(2), rel=1e-5) and d["rect"]["size"][1] == pytest.approx(R * np.sqrt(2), rel=1e-5)


def test_multi_component_area_filter_and_order():
    m = np.zeros((256, 256), dtype=bool)
    m[20:60, 30:90] = True                     # 40 x 60 = 2400, minus the 4 corner pixels the opening removes (label 1)
    m[100:180, 100:200] = True                 # 80 x 100 = 8000 - 4 (label 2)
    m[200:210, 10:25] = True                   # 150 px < 200: dropped
    m[230, 100:140] = True                     # 1-pixel line: removed by the opening
    clean, dets = G.postprocess_minarearect_multi(_prob_from_mask(m))
    assert [d["label"] for d in dets] == [2, 1] and [d["area"] for d in dets] == [7996, 2396]      # sorted by area, ids in raster order
    assert clean[205, 15] == 0 and clean[230, 120] == 0 and clean[30, 40] == 255
    assert G.min_area(256, 256) == 200 and G.min_area(2048, 3072) == 5033


def test_rotated_rectangle_matches_its_construction():
    """A 30-degree rectangle of 90 x 50: the minimum-area rectangle of its pixel set must come back with sides and angle close to
    the construction (pixelisation: within 1.5 px and 1.5 degrees) and diagonals close to hypot(90, 50)."""
    yy, xx = np.mgrid[0:200, 0:200].astype(np.float64)
    th = np.deg2rad(30.0)
    u = (xx - 100) * np.cos(th) + (yy - 100) * np.sin(th)
    v = -(xx - 100) * np.sin(th) + (yy - 100) * np.cos(th)
    m = (np.abs(u) <= 45) & (np.abs(v) <= 25)
    _, dets = G.postprocess_minarearect_multi(_prob_from_mask(m))
    d = dets[0]
    sides = sorted(float(s) for s in d["rect"]["size"])
    assert abs(sides[1] - 90) <= 1.5 and abs(sides[0] - 50) <= 1.5
    ux, uy = (float(c) for c in d["rect"]["u"])
    ang = np.rad2deg(np.arctan2(uy, ux)) % 90.0
    assert min(abs(ang - 30.0), abs(ang - 30.0 + 90), abs(ang - 30 - 90)) <= 1.5 or abs((ang % 90) - 30) <= 1.5
    assert abs(d["d_mean"] - np.hypot(90, 50)) <= 2.5
    assert d["center"] == pytest.approx((100.0, 100.0), abs=0.75)
