"""Device 4-vertex fit (-m gpu): vk_geom_quadrilateral through the host mirror of ui_infer_quadrilateral.py:423-530 against
oracle/quad_oracle.py on the same probability maps.  Everything it reports is compared exactly: clean mask, labels, areas, border
point count, hull size, branch, candidate count, int32 corners, centre, float64 quality and diagonals."""
import importlib
import math

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import quad_oracle as Q
from test_geometry_gpu import _diamonds

pytestmark = pytest.mark.gpu
vk = importlib.import_module("vickers-hardness-unet_amd")
DEV = torch.device("cuda:0")


def _check(prob, dets_g, clean_g, **kw):
    clean_o, dets_o = Q.postprocess_quadrilateral_multi(prob, **kw)
    assert np.array_equal(clean_g, clean_o)
    assert [d["label"] for d in dets_g] == [d["label"] for d in dets_o], ([d["label"] for d in dets_g], [d["label"] for d in dets_o])
    for dg, do in zip(dets_g, dets_o):
        tag = (dg["label"], dg["branch"], do["branch"], dg["box"].tolist(), do["box"].tolist())
        assert dg["flags"] == 0, tag
        assert dg["area"] == do["area"]
        assert dg["contour_points"] == len(do["contour"]), (tag, dg["contour_points"], len(do["contour"]))
        assert dg["hull_vertices"] == len(do["hull"]), tag
        assert dg["branch"] == do["branch"] and dg["n_candidates"] == do["n_candidates"], tag
        assert np.array_equal(dg["box"], do["box"]), tag
        assert dg["center"] == do["center"], tag
        assert dg["quality"] == Q.quad_quality(Q.robust_quadrilateral_from_contour(do["contour"])), tag
        assert (dg["d1"], dg["d2"], dg["d_mean"]) == (do["d1"], do["d2"], do["d_mean"]), tag
    return dets_o


@pytest.mark.parametrize("h,w,seed", [(512, 512, 1), (512, 512, 2), (300, 420, 3), (1024, 1280, 4), (64, 64, 5), (2048, 3072, 6)])
def test_quadrilateral_matches_oracle_on_noisy_diamonds(h, w, seed):
    prob = _diamonds(h, w, seed)
    clean, dets = vk.postprocess_quadrilateral_multi(None, prob)
    assert clean.shape == (h, w) and clean.dtype == np.uint8
    ref = _check(prob, dets, clean)
    if min(h, w) >= 300:
        assert len(ref) >= 1


@pytest.mark.parametrize("outset", [0, 1, 2, 3])
def test_fit_outset_values(outset):
    prob = _diamonds(384, 384, 21, n=2)
    clean, dets = vk.postprocess_quadrilateral_multi(None, prob, fit_outset_px=outset)
    _check(prob, dets, clean, fit_outset_px=outset)


def test_real_label_masks():
    """The reference's own hand-labelled masks (tests/golden/real_masks.npz, data only) through both implementations."""
    z = np.load(GOLDEN / "real_masks.npz")
    for i, nm in enumerate(z["names"]):
        h, w = (int(v) for v in z[f"shape_{i}"])
        mask = np.unpackbits(z[f"bits_{i}"], axis=1)[:, :w].astype(np.float32).reshape(h, w)
        clean, dets = vk.postprocess_quadrilateral_multi(None, mask)
        ref = _check(mask, dets, clean)
        assert len(ref) >= 1, nm
        # and the rectangle path on the same real shapes
        from oracle import geometry_oracle as G
        c2, d2 = vk.postprocess_minarearect_multi(None, mask)
        co, do = G.postprocess_minarearect_multi(mask)
        assert np.array_equal(c2, co) and [d["box"].tolist() for d in d2] == [d["box"].tolist() for d in do], nm


def _polygon(S, pts):
    yy, xx = np.mgrid[0:S, 0:S].astype(np.float64)
    cx, cy = np.mean(pts, axis=0)
    m = np.ones((S, S), bool)
    for i in range(len(pts)):
        a, b = pts[i], pts[(i + 1) % len(pts)]
        cr = (b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])
        cc = (b[0] - a[0]) * (cy - a[1]) - (b[1] - a[1]) * (cx - a[0])
        m &= (cr * np.sign(cc) >= 0)
    return m


def test_fallback_branches_and_odd_shapes():
    """Pentagons (sub-sampling branch), triangles, hexagons, slivers, non-convex shapes (plus, L), a one-pixel-high line after opening,
    components touching the map border: the control flow the noisy diamonds never reach."""
    S = 160
    shapes = []
    for n in (3, 5, 6, 8):
        for r in (20, 35, 60):
            for ph in (0.0, 0.2, 0.5):
                for ax in (1.0, 0.4):
                    pts = [(80 + ax * r * math.cos(ph + 2 * math.pi * k / n), 80 + r * math.sin(ph + 2 * math.pi * k / n)) for k in range(n)]
                    shapes.append((f"reg{n}_r{r}_ph{ph}_ax{ax}", _polygon(S, pts)))
    plus = np.zeros((S, S), bool); plus[70:90, 20:140] = True; plus[20:140, 70:90] = True
    ell = np.zeros((S, S), bool); ell[20:140, 20:40] = True; ell[120:140, 20:140] = True
    line = np.zeros((S, S), bool); line[78:82, 10:150] = True
    edge = np.zeros((S, S), bool); edge[0:30, 0:50] = True; edge[130:160, 100:160] = True        # touching two borders each
    ring = _polygon(S, [(80 + 60 * math.cos(t), 80 + 60 * math.sin(t)) for t in np.linspace(0, 2 * math.pi, 40, endpoint=False)])
    ring &= ~_polygon(S, [(80 + 40 * math.cos(t), 80 + 40 * math.sin(t)) for t in np.linspace(0, 2 * math.pi, 40, endpoint=False)])
    shapes += [("plus", plus), ("L", ell), ("line", line), ("edge", edge), ("ring", ring)]
    branches = set()
    for name, m in shapes:
        prob = m.astype(np.float32)
        clean, dets = vk.postprocess_quadrilateral_multi(None, prob)
        ref = _check(prob, dets, clean)
        branches |= {d["branch"] for d in ref}
    assert {"bisection", "subsample"} <= branches, branches


def test_batch_device_input_and_empty_map():
    probs = np.stack([_diamonds(256, 384, s, n=2) for s in (11, 12, 13, 14)])
    probs[3] = 0.1
    t = torch.from_numpy(probs).to(DEV)
    clean, dets = vk.postprocess_quadrilateral_batch(t)
    assert clean.is_cuda and clean.shape == (4, 256, 384) and dets[3] == []
    for b in range(4):
        _check(probs[b], dets[b], clean[b].cpu().numpy())
    again_clean, again = vk.postprocess_quadrilateral_batch(t)
    assert torch.equal(clean, again_clean)
    assert [[d["box"].tolist() for d in ds] for ds in again] == [[d["box"].tolist() for d in ds] for ds in dets]
    with pytest.raises(vk.VkError):
        vk.postprocess_quadrilateral_batch(torch.zeros(1, 8, 8))
    with pytest.raises(vk.VkError):
        vk.postprocess_quadrilateral_batch(t, fit_outset_px=4)


def test_many_components_and_list_capacity():
    """A grid of 30 small squares (every one above the area threshold of a 512x512 map) with max_components 8: the clean mask keeps
    all of them, the list holds the first eight in label order."""
    prob = np.zeros((512, 512), np.float32)
    for i in range(5):
        for j in range(6):
            prob[20 + i * 95:20 + i * 95 + 30 + 2 * i, 15 + j * 80:15 + j * 80 + 28 + j] = 1.0
    clean, dets = vk.postprocess_quadrilateral_batch(torch.from_numpy(prob[None]).to(DEV), max_components=8)
    clean_o, dets_o = Q.postprocess_quadrilateral_multi(prob)
    assert np.array_equal(clean[0].cpu().numpy(), clean_o) and len(dets_o) == 30 and len(dets[0]) == 8
    by_label = {d["label"]: d for d in dets_o}
    for d in dets[0]:
        assert d["label"] <= 8 and np.array_equal(d["box"], by_label[d["label"]]["box"])
