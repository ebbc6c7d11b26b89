"""SURVEY.md section 8a row A10 through the C-ABI on the GPU box: the product's initialisation helpers against the oracle.
  hcmvs_splat_init        vs oracle/hcmvs_oracle.c hcor_splat_init   (SceneDensify.cpp:783-808)      -> bit-exact
  hcmvs_triangulate_init  vs oracle/triangulate_init.py (Qhull)      (DepthMap.cpp:1796-1936)        -> 1e-4 relative
and the estimate started from either side's maps gives the same bits (the GPU estimate never sees which side made them)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import triangulate_init as TO  # noqa: E402

pytestmark = pytest.mark.gpu
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")


@pytest.fixture(scope="module")
def ctx():
    c = binding.Context(0)
    yield c
    c.close()


def oracle_splat(view, pts):
    h, w = view["gray"].shape
    ref = O.make_view(view)
    d = np.zeros((h, w), np.float32); n = np.full((h, w, 3), 7.0, np.float32)   # the splat only zeroes the normals it touches
    lo = C.c_float(); hi = C.c_float()
    O.lib().hcor_splat_init(C.byref(ref), O.fptr(np.ascontiguousarray(pts, np.float32)), len(pts), O.fptr(d), O.fptr(n), C.byref(lo), C.byref(hi))
    return d, n, lo.value, hi.value


@pytest.mark.parametrize("w,h,f,n,seed", [(160, 120, 150.0, 300, 1), (97, 131, 120.0, 40, 2), (640, 480, 500.0, 2000, 3)])
def test_splat_init_matches_oracle(ctx, w, h, f, n, seed):
    views = synth.make_views(w, h, f, 1, seed=seed)
    pts = synth.sparse_points(views, n, seed=seed + 10)
    # points that project outside the image / next to the border exercise the clamping of the 5x5 block
    v = views[0]
    extra = []
    for (x, y, z) in ((-1.0, 5.0, 9.0), (w + 0.4, h - 1.0, 11.0), (1.0, 1.0, 8.5), (w - 2.0, 0.0, 12.0)):
        Xc = np.array([(x - v["K"][0, 2]) * z / v["K"][0, 0], (y - v["K"][1, 2]) * z / v["K"][1, 1], z])
        extra.append(Xc @ v["R"] + v["C"])
    pts = np.vstack([pts, np.asarray(extra, np.float32)]).astype(np.float32)
    ctx.upload_view(0, v["gray"], v["K"], v["R"], v["C"])
    d, nm, lo, hi = ctx.splat_init(0, pts)
    od, onm, olo, ohi = oracle_splat(v, pts)
    assert np.array_equal(d, od)
    assert lo == olo and hi == ohi
    assert np.array_equal(nm[d > 0], np.zeros_like(nm[d > 0])) and np.array_equal(onm[od > 0], np.zeros_like(onm[od > 0]))
    assert 0 < (d > 0).mean() < 1


def test_triangulate_init_matches_oracle_through_context(ctx):
    w, h, f = 320, 200, 300.0
    views = synth.make_views(w, h, f, 2, seed=6)
    v = views[0]
    pts = synth.sparse_points(views, 400, seed=9)
    for i, u in enumerate(views):
        ctx.upload_view(i, u["gray"], u["K"], u["R"], u["C"])
    d, nm, lo, hi = ctx.triangulate_init(0, pts)
    od, onm, olo, ohi = TO.triangulate_init(w, h, v["K"], v["R"], v["C"], pts)
    assert lo == pytest.approx(olo, rel=1e-6) and hi == pytest.approx(ohi, rel=1e-6)
    rel = np.abs(d - od) / od
    assert (rel < 1e-4).mean() > 0.999 and np.median(rel) < 1e-6
    assert (np.sum(nm * onm, -1) > 0.9999).mean() > 0.999
    # the estimate from the product's init equals the oracle's estimate from the same maps bit for bit (the init itself
    # is compared above; what follows it is row A0-A9)
    pg = binding.default_params(adapthalfwin=6, n_estimation_iters=2, seed=5)
    po = O.default_params(adapthalfwin=6, n_estimation_iters=2, seed=5, arith_mode=O.ARITH_DEVICE, order=O.ORDER_ROWS, n_threads=8)
    got = ctx.estimate(0, [1, 2], pg, lo, hi, d, nm)
    want = O.estimate(views, po, lo, hi, d, nm)
    for g, wv in zip(got, want[:3]):
        assert np.array_equal(g, wv)
    gt = v["depth"]
    m = got[0] > 0
    assert m.mean() > 0.5 and (np.abs(got[0] - gt)[m] / gt[m] < 0.01).mean() > 0.8
