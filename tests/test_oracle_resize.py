"""Known answers of the oracle's cv::resize restatement (oracle/hcmvs_oracle.c hcor_resize_gray; reference call
DepthData::ViewData::ScaleImage, frame_main/libs/MVS/DepthMap.h:233-238).  OpenCV is absent and the reference holds no
fixture (parity unpinned): the vectors follow from the published INTER_AREA / INTER_CUBIC definitions."""
import numpy as np
import pytest

import oracle_lib as O


def test_area_integer_factor_is_the_block_mean():
    rng = np.random.RandomState(0)
    g = rng.uniform(0, 1, (60, 80)).astype(np.float32)
    a = O.resize_gray(g, 0.5)
    assert a.shape == (30, 40)
    want = ((g[0::2, 0::2] + g[0::2, 1::2]) + g[1::2, 0::2] + g[1::2, 1::2]) * np.float32(0.25)   # rows outer, columns inner
    assert np.array_equal(a, want)
    b = O.resize_gray(g[:, :78], 1.0 / 3.0)                      # 60 x 78 -> 20 x 26, 3 x 3 blocks
    assert b.shape == (20, 26) and np.abs(b - g[:, :78].reshape(20, 3, 26, 3).mean((1, 3))).max() < 1e-6


def test_area_fractional_factor_weights():
    # one row, factor 1.6 (scale 0.625): destination cell d covers source [1.6 d, 1.6 d + 1.6)
    src = np.arange(16, dtype=np.float32).reshape(1, 16).repeat(8, 0)
    out = O.resize_gray(src, 0.625)
    assert out.shape == (5, 10)
    # cell 0: [0, 1.6) = pixel 0 whole + 0.6 of pixel 1, weights 1/1.6 and 0.6/1.6
    assert out[0, 0] == pytest.approx((0 * 1 + 1 * 0.6) / 1.6, rel=1e-6)
    # cell 1: [1.6, 3.2) = 0.4 of pixel 1 + pixel 2 + 0.2 of pixel 3
    assert out[0, 1] == pytest.approx((1 * 0.4 + 2 * 1 + 3 * 0.2) / 1.6, rel=1e-6)
    assert np.allclose(out, out[0][None, :], atol=1e-6)          # rows are identical -> the vertical weights sum to 1
    c = np.full((50, 70), 0.37, np.float32)
    for s in (0.62, 0.8, 0.55):
        r = O.resize_gray(c, s)
        sf = float(np.float32(s))                                 # the scale is a float in the reference (ViewData::scale)
        assert r.shape == (int(np.rint(50 * sf)), int(np.rint(70 * sf))) and np.abs(r - 0.37).max() < 1e-6


def test_cubic_reproduces_linear_ramps_and_interpolates():
    x = np.tile(np.linspace(0, 1, 64, dtype=np.float32), (48, 1))
    r = O.resize_gray(x, 2.0)
    assert r.shape == (96, 128)
    inner = r[10, 8:-8]
    assert np.abs(np.diff(inner) - np.diff(inner).mean()).max() < 2e-3    # a ramp stays a ramp away from the replicated border
    # Keys kernel A = -0.75 at t = 0.25: weights (-0.10546875, 0.87890625, 0.26171875, -0.03515625); scale 2 -> fx = x/2 - 0.25
    imp = np.zeros((9, 9), np.float32); imp[4, 4] = 1
    out = O.resize_gray(imp, 2.0)
    # destination 9 has fx = 4.25 -> taps 3..6 with t = 0.25: the impulse at 4 gets weight w1
    assert out[9, 9] == pytest.approx(0.87890625 ** 2, rel=1e-6)
    assert out[9, 7] == pytest.approx(0.87890625 * 0.26171875, rel=1e-6)   # destination 7: fx = 3.25 -> taps 2..5, the impulse is tap 2 (w2)
    assert out[9, 5] == pytest.approx(0.87890625 * -0.03515625, rel=1e-5)  # destination 5: fx = 2.25 -> taps 1..4, the impulse is tap 3 (w3)
    assert O.resize_gray(np.full((20, 30), 0.6, np.float32), 1.7).shape == (34, 51)


def test_area_enlarging_is_bilinear_with_overlap_weights():
    """cv::resize INTER_AREA with a destination larger than the source (restore/libs/MVS/SceneDensify.cpp:523-524, the `restore`
    variant's up-sampling of the previous level's maps): a destination pixel that lies inside one source pixel copies it, one that
    straddles two mixes them by the overlap (OpenCV 4.2 resize.cpp, area mode of the bilinear kernel)."""
    src = np.array([[0, 10, 20, 30]], np.float32).repeat(3, 0)
    out = O.resize_area_up(src, 8, 6)                       # exact factor 2: every destination pixel lies inside one source pixel
    assert np.array_equal(out, np.repeat(np.repeat(src, 2, 0), 2, 1))
    out = O.resize_area_up(src, 6, 3)                       # factor 1.5: destination x covers source [x / 1.5, (x + 1) / 1.5)
    #   x = 1 covers [0.667, 1.333): one third in pixel 0, two thirds... OpenCV's weight of the RIGHT pixel is (x + 1) - (sx + 1) * 1.5 = 0.5
    assert np.allclose(out[0], [0, 5, 10, 20, 25, 30], atol=1e-5)
    rng = np.random.RandomState(3)
    m = rng.uniform(1, 9, (13, 17, 3)).astype(np.float32)
    up = O.resize_area_up(m, 40, 31)
    assert up.shape == (31, 40, 3) and up.min() >= m.min() - 1e-5 and up.max() <= m.max() + 1e-5      # a convex mix of two neighbours
    assert np.array_equal(up[0, 0], m[0, 0]) and np.array_equal(up[-1, -1], m[-1, -1])


def test_c_abi_host_helper_equals_the_restatement():
    """hcmvs_resize_area_up (host helper of the C-ABI, table-driven like OpenCV) against the oracle's per-pixel restatement: bit
    for bit, one and three channels, integer and fractional factors, down to a 1-pixel source"""
    import importlib
    binding = importlib.import_module("hc-mvs_amd.binding")
    rng = np.random.RandomState(5)
    for (sh, sw, dh, dw) in [(48, 64, 96, 128), (45, 60, 90, 121), (30, 40, 77, 101), (1, 1, 5, 7), (16, 16, 16, 16), (7, 5, 8, 5)]:
        for ch in (1, 3):
            m = rng.uniform(0, 12, (sh, sw) if ch == 1 else (sh, sw, ch)).astype(np.float32)
            m[rng.uniform(size=m.shape[:2]) < 0.1] = 0      # holes, as a depth map has them
            assert np.array_equal(binding.resize_area_up(m, dw, dh), O.resize_area_up(m, dw, dh)), (sh, sw, dh, dw, ch)
    with pytest.raises(binding.HcmvsError):
        binding.resize_area_up(np.zeros((8, 8), np.float32), 4, 8)   # shrinking is not this entry's job
