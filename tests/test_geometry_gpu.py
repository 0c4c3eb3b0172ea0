"""Device geometry post-processing (-m gpu): vk_geom_minarearect through the host mirror of ui_infer_rectangle.py:291-381 against
oracle/geometry_oracle.py on the same probability maps.  Integer work (clean mask, label ids, areas, hull size, int32 corners) must
be bit-exact; the float32 rectangle parameters are computed in the same operation order and compared exactly as well (tolerance
1e-6 relative where stated); diagonals are float64 of exact integers."""
import importlib

import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import geometry_oracle as G

pytestmark = pytest.mark.gpu
vk = importlib.import_module("vickers-hardness-unet_amd")
DEV = torch.device("cuda:0")


def _diamonds(h, w, seed, n=3, noise=True):
    """Probability map with n rotated squares (the shape of a Vickers indentation), soft edges, speckle and sub-threshold blobs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    p = np.full((h, w), 0.05)
    for k in range(n):
        cx, cy = rng.uniform(0.15, 0.85) * w, rng.uniform(0.15, 0.85) * h
        half = rng.uniform(0.04, 0.16) * min(h, w)
        th = rng.uniform(0, np.pi / 2)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        d = np.maximum(np.abs(u), np.abs(v)) - half
        p = np.maximum(p, 1.0 / (1.0 + np.exp(d / 1.5)))
    if noise:
        p = p + rng.normal(scale=0.08, size=p.shape)                     # ragged borders, pin holes, specks
        for _ in range(6):                                               # small blobs below the area threshold
            by, bx = rng.integers(5, h - 5), rng.integers(5, w - 5)
            p[by - 3:by + 3, bx - 4:bx + 4] = 0.95
    return np.clip(p, 0.0, 1.0).astype(np.float32)


def _compare(prob, dets_g, clean_g, **kw):
    clean_o, dets_o = G.postprocess_minarearect_multi(prob, **kw)
    assert np.array_equal(clean_g, clean_o)
    assert [d["label"] for d in dets_g] == [d["label"] for d in dets_o]
    for dg, do in zip(dets_g, dets_o):
        assert dg["area"] == do["area"]
        assert dg["hull_vertices"] == len(do["hull"]), (dg["label"], dg["hull_vertices"], len(do["hull"]))
        assert np.array_equal(dg["box"], do["box"]), (dg["box"], do["box"], do["rect"])
        assert dg["center"] == pytest.approx(do["center"], rel=1e-6, abs=1e-4)
        assert dg["size"] == pytest.approx(tuple(float(s) for s in do["rect"]["size"]), rel=1e-6)
        assert (dg["d1"], dg["d2"], dg["d_mean"]) == (do["d1"], do["d2"], do["d_mean"])
    return dets_o


@pytest.mark.parametrize("h,w,seed", [(512, 512, 1), (512, 512, 2), (300, 420, 3), (1024, 1280, 4), (64, 64, 5), (2048, 3072, 6)])
@pytest.mark.parametrize("thresh", [0.5, 0.45])
def test_minarearect_matches_oracle(h, w, seed, thresh):
    if h * w > 4_000_000 and thresh != 0.5:
        pytest.skip("one full-size (3072x2048 micrograph) case is enough")
    prob = _diamonds(h, w, seed)
    clean, dets = vk.postprocess_minarearect_multi(None, prob, bin_thresh=thresh)
    assert clean.shape == (h, w) and clean.dtype == np.uint8
    ref = _compare(prob, dets, clean, bin_thresh=thresh)
    if min(h, w) >= 300:
        assert len(ref) >= 1


def test_batch_and_device_tensor_input():
    probs = np.stack([_diamonds(256, 384, s, n=2) for s in (11, 12, 13, 14)])
    probs[3] = 0.1                                   # an empty map
    t = torch.from_numpy(probs).to(DEV)
    clean, dets = vk.postprocess_minarearect_batch(t)
    assert clean.is_cuda and clean.shape == (4, 256, 384)
    for b in range(4):
        _compare(probs[b], dets[b], clean[b].cpu().numpy())
    assert dets[3] == [] and int(clean[3].max()) == 0


@pytest.mark.parametrize("k,oi,ci", [(3, 1, 1), (5, 1, 1), (3, 2, 2), (3, 0, 0), (1, 1, 1), (7, 1, 1)])
def test_morphology_variants_and_every_component_kept(k, oi, ci):
    """min_area 1 keeps every component: clean == thresholded + open/close mask, ids and areas of all (hundreds of) components."""
    rng = np.random.default_rng(7 + k)
    prob = np.clip(ndimage.gaussian_filter(rng.normal(size=(200, 333)), 1.2) * 3 + 0.45, 0, 1).astype(np.float32)
    t = torch.from_numpy(prob[None]).to(DEV)
    clean, dets = vk.postprocess_minarearect_batch(t, bin_thresh=0.5, min_area_frac=0.0, morph_kernel=k, open_iter=oi, close_iter=ci,
                                                   max_components=4096)
    mask = G.open_close(G.binarize(prob, 0.5), k, oi, ci) if k > 1 else G.binarize(prob, 0.5)
    labels, areas = G.label8(mask)
    # the reference's floor of 200 px applies (max(200, ...)): compare the clean mask with the components of >= 200 px
    keep = np.isin(labels, [i for i in range(1, len(areas)) if areas[i] >= 200])
    assert np.array_equal(clean[0].cpu().numpy() > 0, keep)
    assert sorted(d["label"] for d in dets[0]) == [i for i in range(1, len(areas)) if areas[i] >= 200]
    for d in dets[0]:
        assert d["area"] == int(areas[d["label"]])


def test_touching_border_single_row_and_capacity():
    prob = np.full((96, 160), 0.1, dtype=np.float32)
    prob[0:40, 0:30] = 0.9                 # touches two borders
    prob[60:96, 120:160] = 0.9             # touches the other two
    prob[50, 40:100] = 0.9                 # one-pixel line: removed by the opening
    clean, dets = vk.postprocess_minarearect_multi(None, prob)
    _compare(prob, dets, clean)
    assert len(dets) == 2
    # capacity: more kept components than max_components -> clean still complete, count reported, list truncated in label order
    prob = np.full((128, 512), 0.1, dtype=np.float32)
    for i in range(10):
        prob[20:50, 10 + 50 * i:40 + 50 * i] = 0.9
    t = torch.from_numpy(prob[None]).to(DEV)
    clean, dets = vk.postprocess_minarearect_batch(t, max_components=4)
    clean_o, dets_o = G.postprocess_minarearect_multi(prob)
    assert np.array_equal(clean[0].cpu().numpy(), clean_o) and len(dets_o) == 10
    assert sorted(d["label"] for d in dets[0]) == [1, 2, 3, 4]


def test_argument_errors_and_no_cpu_path():
    with pytest.raises(vk.VkError):
        vk.postprocess_minarearect_batch(torch.zeros(1, 8, 8))
    with pytest.raises(vk.VkError):
        vk.postprocess_minarearect_batch(torch.zeros(1, 8, 8, device=DEV), morph_kernel=4)
    with pytest.raises(vk.VkError):
        vk.postprocess_minarearect_batch(torch.zeros(1, 5000, 8, device=DEV))
