"""The rescaled-neighbour path (reference DepthData::ViewData::ScaleImage, DepthMap.h:233-238,
used at SceneDensify.cpp:372-374): hcmvs_rescale_view on the device against the oracle's cv::resize restatement, bit for
bit, and an estimate whose source views include rescaled ones against the oracle on the same (rescaled) inputs."""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")


@pytest.fixture(scope="module")
def ctx():
    c = binding.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("scale", [0.5, 0.62, 0.8, 0.33, 1.2, 1.7, 2.5, 0.25])
def test_rescale_view_matches_oracle(ctx, scale):
    rng = np.random.RandomState(int(scale * 100))
    h, w = 97, 131
    g = rng.uniform(0, 1, (h, w)).astype(np.float32)
    K = np.array([[120.0, 0, 64.5], [0, 118.0, 47.0], [0, 0, 1]])
    ctx.upload_view(0, g, K, np.eye(3), np.zeros(3))
    got, K2 = ctx.rescale_view(0, 1, scale)
    want = O.resize_gray(g, scale)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    f = max(want.shape[1], want.shape[0]) / max(w, h)           # Image::GetCamera: normalised K times max(w', h')
    assert np.allclose(K2, np.array([[120.0 * f, 0, 64.5 * f], [0, 118.0 * f, 47.0 * f], [0, 0, 1]]), rtol=1e-12)


def test_scale_within_15_percent_is_refused(ctx):
    g = np.zeros((64, 64), np.float32)
    ctx.upload_view(0, g, np.eye(3), np.eye(3), np.zeros(3))
    for s in (1.0, 0.9, 1.14):
        with pytest.raises(binding.HcmvsError):
            ctx.rescale_view(0, 1, s)


def test_estimate_with_rescaled_sources_bit_exact(ctx):
    """two of the three source views are resampled copies (0.7x and 1.4x) with their cameras adjusted like the reference
    does; the estimate against them equals the oracle's on the oracle-resampled images"""
    views = synth.make_views(144, 112, 120.0, 3, seed=23)
    for i, v in enumerate(views):
        ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
    oviews = [views[0], views[1]]
    ids = [1]
    for src, dst, s in ((2, 10, 0.7), (3, 11, 1.4)):
        g, K2 = ctx.rescale_view(src, dst, s)
        og = O.resize_gray(views[src]["gray"], s)
        assert np.array_equal(g, og)
        oviews.append(dict(gray=og, K=K2, R=views[src]["R"], C=views[src]["C"]))
        ids.append(dst)
    pts = synth.sparse_points(views, 100)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    pg = binding.default_params(adapthalfwin=6, n_estimation_iters=3, seed=31)
    po = O.default_params(adapthalfwin=6, n_estimation_iters=3, seed=31, arith_mode=O.ARITH_DEVICE, order=O.ORDER_ROWS, n_threads=8)
    got = ctx.estimate(0, ids, pg, dmin, dmax, d0, n0)
    want = O.estimate(oviews, po, dmin, dmax, d0, n0)
    for a, b in zip(got, want[:3]):
        assert np.array_equal(a, b)
    m = got[0] > 0
    gt = views[0]["depth"]
    assert m.mean() > 0.5 and (np.abs(got[0] - gt)[m] / gt[m] < 0.01).mean() > 0.8
