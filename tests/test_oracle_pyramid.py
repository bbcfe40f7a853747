"""Pins the oracle's pyramid restatement against independent float64 numpy definitions of the OpenCV routines
 and against the reference's structural quirks."""
import numpy as np
import pytest

from oracle import oracle as O


def _reflect101(idx, n):
    idx = np.abs(idx)
    return np.where(idx >= n, 2 * n - 2 - idx, idx)


def blur_f64(img):
    img = img.astype(np.float64)
    r, c = img.shape
    xs = np.arange(c)
    t = 0.5 * img + 0.25 * (img[:, _reflect101(xs - 1, c)] + img[:, _reflect101(xs + 1, c)])
    ys = np.arange(r)
    return 0.5 * t + 0.25 * (t[_reflect101(ys - 1, r), :] + t[_reflect101(ys + 1, r), :])


def pyrdown_f64(img):
    img = img.astype(np.float64)
    r, c = img.shape
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    xs = 2 * np.arange(c // 2)
    h = sum(k[i] * img[:, _reflect101(xs - 2 + i, c)] for i in range(5))
    ys = 2 * np.arange(r // 2)
    return sum(k[i] * h[_reflect101(ys - 2 + i, r), :] for i in range(5))


@pytest.mark.parametrize("shape", [(376, 1241), (48, 64), (7, 9)])
def test_blur_exact_for_u8_input(shape):
    img = np.random.default_rng(0).integers(0, 256, shape).astype(np.float32)
    assert np.array_equal(O.blur3x3(img).astype(np.float64), blur_f64(img))  # multiples of 1/16 < 256: exact in fp32


def test_blur_constant_and_impulse():
    assert np.array_equal(O.blur3x3(np.full((9, 11), 7.0, np.float32)), np.full((9, 11), 7.0, np.float32))
    imp = np.zeros((9, 9), np.float32)
    imp[4, 4] = 16
    out = O.blur3x3(imp)
    assert np.array_equal(out[3:6, 3:6], np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.float32))
    corner = np.zeros((5, 5), np.float32)
    corner[0, 0] = 16
    # reflect-101: row -1 mirrors row 1, so the corner keeps only its own 1/2 * 1/2 weight
    assert O.blur3x3(corner)[0, 0] == 4.0


@pytest.mark.parametrize("shape", [(376, 1241), (188, 620), (94, 310), (10, 13)])
def test_pyrdown_matches_definition(shape):
    img = np.random.default_rng(1).integers(0, 256, shape).astype(np.float32)
    out = O.pyrdown(img)
    assert out.shape == (shape[0] // 2, shape[1] // 2)
    assert np.array_equal(out.astype(np.float64), pyrdown_f64(img))  # u8 input: 16 bits needed, exact


def test_pyramid_levels_and_quirks():
    img = np.random.default_rng(2).integers(0, 256, (376, 1241)).astype(np.float32)
    pyr = O.image_pyramid(img, 4, True)
    assert [p.shape for p in pyr] == [(376, 1241), (188, 620), (94, 310), (47, 155)]
    assert np.array_equal(pyr[0], O.blur3x3(img))
    # L1 comes from the UNSMOOTHED input (ref: src/image_processing_global.cpp:38), not from L0
    assert np.array_equal(pyr[1], O.pyrdown(img))
    assert not np.array_equal(pyr[1], O.pyrdown(pyr[0]))
    assert np.array_equal(pyr[2], O.pyrdown(pyr[1]))
    assert np.array_equal(pyr[3], O.pyrdown(pyr[2]))
    # L1, L2 exact for u8-origin input; L3 needs 32 bits -> within a few ulp of the float64 definition
    assert np.array_equal(pyr[2].astype(np.float64), pyrdown_f64(pyrdown_f64(img)))
    l3 = pyrdown_f64(pyrdown_f64(pyrdown_f64(img)))
    assert np.all(np.abs(pyr[3] - l3) <= 4 * np.spacing(l3.astype(np.float32)))  # a few fp32 roundings inside L3
    nosmooth = O.image_pyramid(img, 4, False)
    assert np.array_equal(nosmooth[0], img) and np.array_equal(nosmooth[1], pyr[1])


def test_depth_pyramid_odd_decimation():
    dep = np.random.default_rng(3).random((376, 1241)).astype(np.float32)
    pyr = O.depth_pyramid(dep, 4)
    assert [p.shape for p in pyr] == [(376, 1241), (188, 620), (94, 310), (47, 155)]
    assert np.array_equal(pyr[0], dep)
    for l in range(1, 4):
        r, c = pyr[l].shape
        assert np.array_equal(pyr[l], pyr[l - 1][1:2 * r:2, 1:2 * c:2])


def test_principal_point_rule():
    # cx = 607.1928 / 304.3464 / 152.9232 / 77.2116, cy = 185.2157 / 93.35785 / 47.428925 / 24.4644625
    cx = [O.lib().orc_cx_level(607.1928, l) for l in range(4)]
    cy = [O.lib().orc_cx_level(185.2157, l) for l in range(4)]
    np.testing.assert_allclose(cx, [607.1928, 304.3464, 152.9232, 77.2116], rtol=2e-7)
    np.testing.assert_allclose(cy, [185.2157, 93.35785, 47.428925, 24.4644625], rtol=2e-7)


def test_median3x3_and_smoothed_depth_pyramid():
    """DepthPyramid(smooth = true): cv::medianBlur(.., 3) (ref: src/image_processing_global.cpp:76-80) restated as the median of
    the 3x3 neighbourhood with a replicated border, against scipy's; the levels below are decimated from the filtered level 0."""
    from scipy import ndimage
    rng = np.random.default_rng(5)
    for shape in ((7, 9), (48, 64), (33, 41)):
        a = rng.random(shape, np.float32)
        a[rng.random(shape) < 0.3] = 0.0          # invalid depth holes
        got = O.median3x3(a)
        assert np.array_equal(got, ndimage.median_filter(a, size=3, mode="nearest"))
    a = rng.random((48, 64), np.float32)
    lv = O.depth_pyramid(a, 4, smooth=True)
    m = O.median3x3(a)
    assert np.array_equal(lv[0], m)
    for k in range(1, 4):
        s = 2 ** k
        assert np.array_equal(lv[k], m[s - 1::s, s - 1::s][:lv[k].shape[0], :lv[k].shape[1]])
    assert np.array_equal(O.depth_pyramid(a, 4)[0], a)      # smooth = false: a plain copy, as before
