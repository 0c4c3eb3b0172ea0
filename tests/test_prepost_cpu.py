"""CPU checks of the pre/post-processing restatement (oracle/prepost_oracle.py) and of the host-side geometry
(SURVEY.md §8(f) rank 1).  cv2 is not importable here, so the resize arithmetic is "parity unpinned"; these tests
pin the restatement to values worked out by hand from OpenCV's published formulae and pin the conventions
(scale rule, rounding, padding position) to the reference's lines."""
import importlib

import numpy as np
import pytest

from oracle import prepost_oracle as P


def test_linear_u8_known_answer():
    # [0, 255] -> width 4: scale 0.5; fx = -0.25 (clamped), 0.25, 0.75, 1.25 (clamped)
    #   dx=1: a = (1536, 512): 255*512 = 130560; ((2048*(130560>>4))>>16) = 255; (255+2)>>2 = 64
    #   dx=2: a = (512, 1536): 255*1536 = 391680; >>4 = 24480; *2048>>16 = 765; (765+2)>>2 = 191
    src = np.array([[0, 255]], dtype=np.uint8)
    assert P.resize_linear_u8(src, 4, 1).tolist() == [[0, 64, 191, 255]]
    # vertical direction, 3 channels
    src3 = np.stack([src.T] * 3, axis=-1)
    assert P.resize_linear_u8(src3, 1, 4)[:, 0, 1].tolist() == [0, 64, 191, 255]


def test_linear_u8_downscale_known_answer():
    # width 4 -> 2: scale 2; fx = 0.5, 2.5 -> samples (0,1) and (2,3) with weights 1024/1024
    src = np.array([[10, 20, 30, 41]], dtype=np.uint8)
    # (10*1024 + 20*1024) = 30720 -> >>4 = 1920 -> *2048>>16 = 60 -> (60+2)>>2 = 15 ; (30+41)*1024 = 72704 -> 4544 -> 142 -> 36
    assert P.resize_linear_u8(src, 2, 1).tolist() == [[15, 36]]


def test_nearest_known_answer():
    src = np.arange(3, dtype=np.uint8)[None, :]
    assert P.resize_nearest(src, 5, 1).tolist() == [[0, 0, 1, 1, 2]]      # floor(x * 0.6)
    assert P.resize_nearest(src, 2, 1).tolist() == [[0, 1]]               # floor(x * 1.5)


def test_same_size_is_a_copy():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_linear_u8(a, 9, 7), a)
    f = rng.random((7, 9), dtype=np.float32)
    assert np.array_equal(P.resize_linear_f32(f, 9, 7), f)
    assert np.array_equal(P.resize_nearest(a[:, :, 0], 9, 7), a[:, :, 0])


def test_linear_f32_keeps_a_ramp_inside_the_image():
    x = np.arange(8, dtype=np.float32)[None, :].repeat(3, 0)
    up = P.resize_linear_f32(x, 16, 6)
    want = np.clip((np.arange(16) + 0.5) * 0.5 - 0.5, 0, 7)
    assert np.allclose(up, want[None, :], atol=1e-6)


@pytest.mark.parametrize("h,w,size,want", [
    (1200, 1600, 512, (0.32, 384, 512, 0, 0)),           # infer_pth_gui.py:18-20
    (300, 400, 512, (1.28, 384, 512, 0, 0)),             # this convention enlarges
    (1001, 333, 512, (512 / 1001, 512, 170, 0, 0)),
])
def test_geometry_pad_br(h, w, size, want):
    assert P.geometry_pad_br(h, w, size) == pytest.approx(want)


@pytest.mark.parametrize("h,w,size,want", [
    (1200, 1600, 512, (0.32, 384, 512, 64, 0)),          # ui_infer_quadrilateral.py:205-212
    (300, 400, 512, (1.0, 300, 400, 106, 56)),           # never enlarges
    (1001, 333, 512, (512 / 1001, 512, 170, 0, 171)),
])
def test_geometry_centered(h, w, size, want):
    assert P.geometry_centered(h, w, size) == pytest.approx(want)


@pytest.mark.parametrize("h,w,size,want", [
    (1200, 1600, 512, (0.32, 384, 512, 64, 0)),          # train.py:70-75 on a 1600x1200 micrograph
    (300, 400, 512, (1.28, 384, 512, 64, 0)),            # LongestMaxSize enlarges (unlike the Qt wrappers' letterbox)
    (1001, 333, 512, (512 / 1001, 512, 170, 0, 171)),
    (2048, 3072, 512, (1 / 6, 341, 512, 85, 0)),         # the dataset's other size: 3072x2048 -> 512x341, 85 rows above, 86 below
])
def test_geometry_train(h, w, size, want):
    assert P.geometry_train(h, w, size) == pytest.approx(want)


def test_host_geometry_matches_oracle():
    vk = importlib.import_module("vickers-hardness-unet_amd")
    for h, w in [(1200, 1600), (300, 400), (1001, 333), (512, 512), (37, 2048), (1, 1)]:
        for size in (256, 512):
            assert vk.prepost.letterbox_geometry(h, w, size, "pad_br") == P.geometry_pad_br(h, w, size)
            assert vk.prepost.letterbox_geometry(h, w, size, "centered") == P.geometry_centered(h, w, size)
            assert vk.prepost.letterbox_geometry(h, w, size, "train") == P.geometry_train(h, w, size)
    with pytest.raises(ValueError):
        vk.prepost.letterbox_geometry(10, 10, 512, "stretch")


def test_preprocess_layout_and_constants():
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 255, 128, 0          # B, G, R
    x, (nh, nw, top, left) = P.preprocess(img, 8, "centered")
    assert x.shape == (3, 8, 8) and x.dtype == np.float32 and (nh, nw, top, left) == (4, 4, 2, 2)
    inside = x[:, 2:6, 2:6]
    want = (np.array([0, 128, 255], dtype=np.float32) / np.float32(255) - P.MEAN) / P.STD      # R, G, B planes
    assert np.array_equal(inside, np.broadcast_to(want[:, None, None], inside.shape))
    border = (np.float32(0) - P.MEAN) / P.STD
    assert np.array_equal(x[:, 0, 0], border)


def test_postprocess_roundtrip_shapes():
    rng = np.random.default_rng(1)
    lg = rng.normal(size=(64, 64)).astype(np.float32) * 4
    for conv in ("pad_br", "centered", "train"):
        for (h, w) in [(100, 150), (40, 30), (64, 64)]:
            geo = P.GEOMETRY[conv](h, w, 64)
            m = P.postprocess_mask(lg, *geo[1:], (h, w))
            pr = P.postprocess_prob(lg, *geo[1:], (h, w))
            assert m.shape == (h, w) and m.dtype == np.uint8 and set(np.unique(m)) <= {0, 255}
            assert pr.shape == (h, w) and pr.dtype == np.float32 and pr.min() >= 0 and pr.max() <= 1
