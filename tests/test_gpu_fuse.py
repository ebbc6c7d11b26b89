"""GPU parity of FilterDepthMap / FuseDepthMaps against the CPU oracle through the C-ABI: bit-exact, including
the order of the fused points and the depth estimates fusion invalidates."""
import importlib

import numpy as np
import pytest

import oracle_lib as O
from fusion_scene import make_maps

pytestmark = pytest.mark.gpu
binding = importlib.import_module("hc-mvs_amd.binding")


@pytest.fixture(scope="module")
def ctx():
    c = binding.Context(0)
    yield c
    c.close()


def upload(ctx, maps):
    for i, m in enumerate(maps):
        ctx.upload_view(i, m["gray"], m["K"], m["R"], m["C"], bgr=m["bgr"])
        ctx.set_depthmap(i, m["depth"], m["normal"], m["conf"], m["d_min"], m["d_max"])
        ctx.set_neighbors(i, m["neighbors"])


@pytest.mark.parametrize("kw", [dict(noise=0.0), dict(noise=0.003, outliers=0.05, holes=0.1),
                                dict(w=144, h=112, f=130.0, n_views=7, noise=0.002, outliers=0.03)])
def test_fuse_bit_exact(ctx, kw):
    maps, order = make_maps(**kw)
    upload(ctx, maps)
    want = O.fuse_depthmaps(maps, order, 200000)
    got = ctx.fuse(order, 200000)
    assert got["n_points"] == want["n_points"] > 100 and got["n_depths"] == want["n_depths"]
    assert np.array_equal(got["n_views"], want["n_views"])
    assert np.array_equal(got["xyz"], want["xyz"])          # same points in the same order
    assert np.array_equal(got["normal"], want["normal"]) and np.array_equal(got["bgr"], want["bgr"])
    for i, d in enumerate(want["depths"]):                   # the same estimates were invalidated
        assert np.array_equal(ctx.get_depthmap(i)[0], d)


def test_fuse_scale_mismatched_neighbour(ctx):
    """a neighbour seen at 2.8x the footprint scale: ~8 pixels of every other image land on each of its pixels, so the
    per-pixel link lists are far longer than for equal scales (the round-1 build returned HCMVS_ERR_CAPACITY here; the
    reference's FuseDepthMaps, SceneDensify.cpp:3265-3495, has no such limit).  Bit-exact against the oracle, both ways."""
    for far in (1, 0):
        maps, order = make_maps(w=160, h=128, f=150.0, n_views=5, noise=0.002, outliers=0.02, far=far, far_factor=2.8)
        upload(ctx, maps)
        want = O.fuse_depthmaps(maps, order, 200000)
        got = ctx.fuse(order, 200000)
        assert got["n_points"] == want["n_points"] > 1000 and got["n_depths"] == want["n_depths"]
        assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["n_views"], want["n_views"])
        assert np.array_equal(got["normal"], want["normal"]) and np.array_equal(got["bgr"], want["bgr"])
        for i, d in enumerate(want["depths"]):
            assert np.array_equal(ctx.get_depthmap(i)[0], d)


def test_fuse_independent_clusters_interleaved(ctx):
    """three clusters of views that share no map (every image's neighbours lie in its own cluster), fused in an interleaved order: the
    cloud keeps the order of the sequential loop (SceneDensify.cpp:3302), every image's points at the offset the counts of the earlier
    images add up to (kept on the device, no host synchronisation between the images).  Bit-exact against the sequential oracle
    including point order, view lists and invalidated depths."""
    clusters = [make_maps(w=128, h=96, f=115.0, n_views=4, seed=21 + 5 * k, noise=0.002, outliers=0.04, holes=0.05)[0] for k in range(3)]
    maps = []
    for k, cl in enumerate(clusters):
        for m in cl:
            m = dict(m)
            m["neighbors"] = [4 * k + j for j in m["neighbors"]]
            maps.append(m)
    order = [0, 4, 8, 5, 1, 9, 2, 10, 6, 11, 7, 3]            # clusters interleaved, not in lock-step
    want = O.fuse_depthmaps(maps, order, 400000)
    vcap = int(sum((m["depth"] != 0).sum() for m in maps))
    for rep in range(2):                                    # twice on one context: the scratch of the first call is reused
        upload(ctx, maps)
        got = ctx.fuse_cloud(order, 400000, vcap)
        assert got["n_points"] == want["n_points"] > 3000 and got["n_depths"] == want["n_depths"]
        assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["n_views"], want["n_views"])
        assert np.array_equal(got["normal"], want["normal"]) and np.array_equal(got["bgr"], want["bgr"])
        assert np.array_equal(got["view_ids"], want["view_ids"]) and np.array_equal(got["view_weights"], want["view_weights"])
        for i, d in enumerate(want["depths"]):
            assert np.array_equal(ctx.get_depthmap(i)[0], d)


def test_fuse_options_and_capacity(ctx):
    maps, order = make_maps(noise=0.002)
    upload(ctx, maps)
    want = O.fuse_depthmaps(maps, order, 200000, n_min_views_fuse=3, thr=0.02, normal_deg=15.0, depthweight=0.7, normalweight=1.3)
    got = ctx.fuse(order, 200000, n_min_views_fuse=3, depth_diff_threshold=0.02, normal_diff_deg=15.0, depthweight=0.7,
                   normalweight=1.3, with_colors=False)
    assert got["n_points"] == want["n_points"] and np.array_equal(got["xyz"], want["xyz"]) and got["bgr"] is None
    upload(ctx, maps)
    with pytest.raises(binding.HcmvsError) as e:
        ctx.fuse(order, 10)
    assert e.value.code == binding.ERR_CAPACITY


@pytest.mark.parametrize("adjust", [True, False])
def test_filter_bit_exact(ctx, adjust):
    maps, _ = make_maps(w=144, h=112, f=130.0, n_views=6, noise=0.002, outliers=0.06, holes=0.05)
    upload(ctx, maps)
    for ref in (0, 3):
        nb = maps[ref]["neighbors"][:4]
        ok, d, c, nproc, ndisc = O.filter_depthmap(maps, ref, nb, adjust=adjust)
        gd, gc, gp, gdisc = ctx.filter(ref, nb, adjust=adjust)
        assert ok == 1 and (gp, gdisc) == (nproc, ndisc) and 0 < ndisc < nproc
        assert np.array_equal(gd, d) and np.array_equal(gc, c)
    with pytest.raises(binding.HcmvsError):
        ctx.filter(0, maps[0]["neighbors"][:1], adjust=adjust, n_min_views=2)


@pytest.mark.parametrize("kw", [dict(noise=0.003, outliers=0.05, holes=0.1), dict(w=144, h=112, f=130.0, n_views=7, noise=0.002, outliers=0.03)])
def test_fuse_hashed_order_within_tolerance(ctx, kw):
    """hcmvs_set_fuse_order(1): the same greedy rule with the pixels of an image visited in a hashed order.  Not the
    reference's cloud bit for bit, but within the tolerance the north star states (point count within 1 %), and
    deterministic."""
    maps, order = make_maps(**kw)
    want = O.fuse_depthmaps(maps, order, 200000)
    try:
        ctx.set_fuse_order(1)
        upload(ctx, maps)
        a = ctx.fuse(order, 200000)
        upload(ctx, maps)
        b = ctx.fuse(order, 200000)
    finally:
        ctx.set_fuse_order(0)
    assert a["n_points"] == b["n_points"] and np.array_equal(a["xyz"], b["xyz"]) and np.array_equal(a["n_views"], b["n_views"])
    assert abs(a["n_points"] - want["n_points"]) <= 0.01 * want["n_points"]
    assert abs(a["n_depths"] - want["n_depths"]) <= 0.01 * want["n_depths"]   # later images see other invalidations
    # the clouds describe the same surface: nearly every point of one has a point of the other within a small distance
    from scipy.spatial import cKDTree
    d, _ = cKDTree(want["xyz"]).query(a["xyz"])
    scale = np.linalg.norm(want["xyz"].max(0) - want["xyz"].min(0))
    assert (d < 2e-3 * scale).mean() > 0.97
    with pytest.raises(binding.HcmvsError):
        ctx.set_fuse_order(2)


@pytest.mark.parametrize("kw", [dict(noise=0.003, outliers=0.05, holes=0.1), dict(w=144, h=112, f=130.0, n_views=7, noise=0.002, outliers=0.03)])
def test_fuse_hashed_order_equals_the_oracle_in_that_order(ctx, kw):
    """hcmvs_set_fuse_order(1) against the sequential oracle visiting the pixels of every image in the same hashed order: the same
    points with the same attributes (the oracle emits them in visiting order, the device in raster order, so both are sorted before
    the comparison) and the same invalidated depths, exactly."""
    maps, order = make_maps(**kw)
    want = O.fuse_depthmaps(maps, order, 200000, pixel_order=1)
    try:
        ctx.set_fuse_order(1)
        upload(ctx, maps)
        got = ctx.fuse(order, 200000)
    finally:
        ctx.set_fuse_order(0)
    assert got["n_points"] == want["n_points"] > 100 and got["n_depths"] == want["n_depths"]
    def rows(c):
        r = np.concatenate([c["xyz"].view(np.uint32).astype(np.uint64), c["normal"].view(np.uint32).astype(np.uint64),
                            c["bgr"].astype(np.uint64), c["n_views"][:, None].astype(np.uint64)], axis=1)
        return r[np.lexsort(r.T[::-1])]
    assert np.array_equal(rows(got), rows(want))
    for i, d in enumerate(want["depths"]):
        assert np.array_equal(ctx.get_depthmap(i)[0], d)


def test_fuse_cloud_views_weights_colors_normals(ctx):
    """the complete PointCloud: view lists + weights of every fused point (PointCloud::pointViews /
    pointWeights, SceneDensify.cpp:3376-3411) bit-exact against the oracle; MVS::EstimatePointColors (DepthMap.cpp:2125-2161)
    bit-exact; MVS::EstimatePointNormals (DepthMap.cpp:2221-2269, CGAL PCA over 16 neighbours -- restated, parity unpinned)
    against a brute-force numpy PCA"""
    maps, order = make_maps(w=144, h=112, f=130.0, n_views=6, noise=0.002, outliers=0.03, holes=0.05)
    upload(ctx, maps)
    want = O.fuse_depthmaps(maps, order, 200000)
    vcap = int(sum((m["depth"] != 0).sum() for m in maps))
    got = ctx.fuse_cloud(order, 200000, vcap)
    assert got["n_points"] == want["n_points"] > 1000
    assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["n_views"], want["n_views"])
    assert np.array_equal(got["view_ids"], want["view_ids"]) and np.array_equal(got["view_weights"], want["view_weights"])
    assert len(got["view_ids"]) == int(got["n_views"].sum())
    off = np.concatenate([[0], np.cumsum(got["n_views"].astype(np.int64))]).astype(np.int64)
    for p in (0, 17, got["n_points"] - 1):                      # ascending image ids inside a point's list
        v = got["view_ids"][off[p]:off[p + 1]]
        assert (np.diff(v.astype(np.int64)) > 0).all()
    # colours from the closest view
    gc = ctx.estimate_point_colors(got["xyz"], got["n_views"], got["view_ids"])
    wc = O.estimate_point_colors(maps, want["xyz"], want["n_views"], want["view_ids"])
    assert np.array_equal(gc, wc)
    assert (gc != 255).any() and np.abs(gc.astype(int) - got["bgr"].astype(int)).mean() < 12   # close to the fusion-time colours
    # normals by PCA over the 16 nearest points, flipped towards the first view
    gn = ctx.estimate_point_normals(got["xyz"], got["n_views"], got["view_ids"], 16)
    from scipy.spatial import cKDTree
    X = got["xyz"].astype(np.float64)
    _, nn = cKDTree(X).query(X, k=16)
    sub = np.arange(0, len(X), 7)
    Pn = X[nn[sub]]
    Pc = Pn - Pn.mean(1, keepdims=True)
    w, v = np.linalg.eigh(np.einsum("nki,nkj->nij", Pc, Pc))
    ref = v[:, :, 0]
    first = got["view_ids"][off[:-1]][sub]
    Cs = np.stack([maps[i]["C"] for i in first])
    flip = ((Cs - X[sub]) * ref).sum(1) < 0
    ref[flip] *= -1
    cosang = np.abs((gn[sub] * ref).sum(1))
    well = (w[:, 1] > 50 * np.maximum(w[:, 0], 1e-30))          # a clear plane: the smallest eigenvalue is well separated
    assert well.mean() > 0.5 and (cosang[well] > 1 - 1e-6).mean() > 0.99
    assert ((gn[sub] * ref).sum(1)[well] > 0).mean() > 0.999    # same orientation
    assert np.abs(np.linalg.norm(gn, axis=1) - 1).max() < 1e-5
    assert ((gn * got["normal"]).sum(1) > 0.8).mean() > 0.7     # and they agree with the normals fusion averaged


@pytest.mark.parametrize("vid", [0, 2])
def test_postfilter_bit_exact(ctx, vid):
    """the fork's RemoveSmallSegments (= a whole fusion pass, then the mask of the pixels that ended up in
    a fused point) + GapInterpolation (rows, then columns) + merge, SceneDensify.cpp:2048-2275, 2280-3001, applied at :3939-3958.
    Depth / normal / confidence of the image and the depths the fusion invalidates in the others: bit for bit against the
    oracle (device-association transcendentals)."""
    maps, order = make_maps(w=144, h=112, f=130.0, n_views=5, noise=0.002, outliers=0.05, holes=0.12)
    for m in maps:
        m["conf"] = np.where(m["depth"] > 0, 1.3 - m["conf"], 0).astype(np.float32)   # between outer iterations conf holds the SCORE
    upload(ctx, maps)
    gra = ctx.gradient_map(vid)
    dd, nd, cd, filled = O.postfilter(maps, vid, gra, order, mode=O.ARITH_DEVICE)
    got_filled = ctx.postfilter(vid, order)
    assert got_filled == filled > 100
    d, n, c = ctx.get_depthmap(vid, with_normal=True)
    assert np.array_equal(d, dd[vid]) and np.array_equal(n, nd) and np.array_equal(c, cd)
    for i in range(len(maps)):
        assert np.array_equal(ctx.get_depthmap(i)[0], dd[i])
    assert (d > 0).sum() > (maps[vid]["depth"] > 0).sum()           # holes were filled
    gt = maps[vid]["gt"]
    newly = (d > 0) & (maps[vid]["depth"] == 0)
    assert (np.abs(d - gt)[newly] / gt[newly] < 0.02).mean() > 0.8   # and the filled values lie on the surface


@pytest.mark.parametrize("pairs", [[(10, 11), (20, 22), (90, 99)], [(20, 21)]])
def test_postfilter_gradient_ratio_literal(ctx, pairs):
    """GapInterpolation's long-gap rule compares the float ratio of the gradient-map values at the gap's ends with the DOUBLE
    literal 0.1 (SceneDensify.cpp:2383-2390 rows, :2713-2720 columns): pairs whose ratio rounds to 0.1f -- (10, 11), (20, 22),
    (90, 99) -- stay open, (20, 21) fills.  The device computes the gradient map itself from the uploaded image."""
    from test_oracle_postfilter import gap_scene
    maps, rows, (x0, x1), img = gap_scene(pairs)
    for i, m in enumerate(maps):
        g8 = img if i == 0 else np.full_like(img, 50)
        m["gray"] = g8.astype(np.float32) / 255.0
        m["bgr"] = np.repeat(g8[..., None], 3, -1).copy()
        m["conf"] = np.where(m["depth"] > 0, 0.4, 0).astype(np.float32)
    upload(ctx, maps)
    gra = ctx.gradient_map(0)
    for y, (g0, g1) in zip(rows, pairs):
        assert (int(gra[y, x0]), int(gra[y, x1])) == (g0, g1)
    dd, nd, cd, filled = O.postfilter(maps, 0, gra, [0, 1], mode=O.ARITH_DEVICE)
    assert ctx.postfilter(0, [0, 1]) == filled
    d, n, c = ctx.get_depthmap(0, with_normal=True)
    assert np.array_equal(d, dd[0]) and np.array_equal(n, nd) and np.array_equal(c, cd)
    for y in rows:
        assert ((d[y, x0 + 1:x1] > 0).all() if pairs == [(20, 21)] else (d[y, x0 + 1:x1] == 0).all())


def test_postfilter_sequence_equals_image_after_image(ctx, capfd, monkeypatch):
    """hcmvs_postfilter_sequence (all passes of a fusion enqueued without host synchronisation, one synchronisation per image)
    against the oracle run image after image, as the reference does over an outer iteration (SceneDensify.cpp:3939-3958): every
    image's fusion sees the maps the images before it left.  Depth, normal, confidence of every image: bit for bit."""
    maps, order = make_maps(w=144, h=112, f=130.0, n_views=5, noise=0.002, outliers=0.05, holes=0.12)
    for m in maps:
        m["conf"] = np.where(m["depth"] > 0, 1.3 - m["conf"], 0).astype(np.float32)
    upload(ctx, maps)
    seq = [3, 0, 4, 1, 2]
    cur = [dict(m) for m in maps]
    total = 0
    for vid in seq:
        dd, nd, cd, filled = O.postfilter(cur, vid, ctx.gradient_map(vid), order, mode=O.ARITH_DEVICE)
        total += filled
        for i in range(len(cur)):
            cur[i]["depth"] = dd[i]
        cur[vid]["normal"] = nd; cur[vid]["conf"] = cd
    # the chain computes every fusion after the first INCREMENTALLY (pf_kernels.hip: only what depends on the pixels the previous image's
    # gap interpolation filled and on the estimates the previous fusion zeroed is evaluated again) ...
    monkeypatch.setenv("HCMVS_FUSE_DEBUG", "1")
    capfd.readouterr()
    assert ctx.postfilter_sequence(seq, order) == total > 500
    assert "fusions computed incrementally" in capfd.readouterr().err
    monkeypatch.delenv("HCMVS_FUSE_DEBUG")
    for i in range(len(maps)):
        d, n, c = ctx.get_depthmap(i, with_normal=True)
        assert np.array_equal(d, cur[i]["depth"]) and np.array_equal(n, cur[i]["normal"]) and np.array_equal(c, cur[i]["conf"]), i
    # ... and with every fusion from scratch (HCMVS_PF_FULL: the path of rounds 1-3, the fall-back when the state does not fit)
    upload(ctx, maps)
    monkeypatch.setenv("HCMVS_PF_FULL", "1")
    assert ctx.postfilter_sequence(seq, order) == total
    monkeypatch.delenv("HCMVS_PF_FULL")
    for i in range(len(maps)):
        d, n, c = ctx.get_depthmap(i, with_normal=True)
        assert np.array_equal(d, cur[i]["depth"]) and np.array_equal(n, cur[i]["normal"]) and np.array_equal(c, cur[i]["conf"]), i
    # the same through single calls ...
    upload(ctx, maps)
    assert sum(ctx.postfilter(v, order) for v in seq) == total
    for i in range(len(maps)):
        assert np.array_equal(ctx.get_depthmap(i)[0], cur[i]["depth"])
    # ... and on a fresh context, cloud fusion first (the per-pass scratch is shared between the two entry points)
    c2 = binding.Context(0)
    upload(c2, maps)
    want = O.fuse_depthmaps(maps, order, 200000)
    got = c2.fuse(order, 200000)
    assert got["n_points"] == want["n_points"] and np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["bgr"], want["bgr"])
    for i, d in enumerate(want["depths"]):
        assert np.array_equal(c2.get_depthmap(i)[0], d)
    upload(c2, maps)
    assert c2.postfilter_sequence(seq, order) == total
    for i in range(len(maps)):
        d, n, c = c2.get_depthmap(i, with_normal=True)
        assert np.array_equal(d, cur[i]["depth"]) and np.array_equal(n, cur[i]["normal"]) and np.array_equal(c, cur[i]["conf"]), i
    c2.close()


def chain_maps(w=512, h=24, depth=5.0):
    """three views of a fronto-parallel plane from ONE camera position: A at full horizontal resolution, B and C at half of it, their
    pixel grids shifted by a quarter pixel either way, so that A's pixels 2j, 2j+1 land on B's pixel j and 2j-1, 2j on C's pixel j.
    With nMinViewsFuse = 3 a pixel of A becomes a point only when BOTH its targets are still free: x = 0 is one (it claims B0 and
    C0), so x = 1 (shares B0) is not, so C1 stays free and x = 2 is one ... -- the answer of every pixel of a row hangs on the answer
    of the pixel before it, over the whole row, and it ALTERNATES: the worst case for an iteration that starts from "every pixel is a
    point" (its changes travel one pixel per step)."""
    f = 300.0
    def view(width, fx, cx):
        K = np.array([[fx, 0, cx], [0, f, (h - 1) / 2.0], [0, 0, 1]], np.float64)
        d = np.full((h, width), depth, np.float32)
        n = np.zeros((h, width, 3), np.float32); n[..., 2] = -1
        g = np.full((h, width), 0.5, np.float32)
        g8 = np.full((h, width), 128, np.uint8)
        return dict(K=K, R=np.eye(3), C=np.zeros(3), gray=g, depth=d, normal=n, conf=np.full((h, width), 0.8, np.float32),
                    bgr=np.stack([g8, g8, g8], -1).copy(), d_min=1.0, d_max=10.0, neighbors=[])
    cxA = (w - 1) / 2.0
    A = view(w, f, cxA)
    B = view(w // 2 + 1, f / 2, cxA / 2 - 0.25)
    Cm = view(w // 2 + 1, f / 2, cxA / 2 + 0.25)
    A["neighbors"] = [1, 2]
    return [A, B, Cm], [0, 1, 2]


def test_fuse_alternating_chain_through_the_settle_loop(ctx):
    """the settle iteration (fuse_kernels.hip) converges in two or three steps on estimated maps; here it needs about as many steps as a
    row has pixels, far beyond the full-grid steps that are enqueued, so the single-workgroup loop that finishes the iteration does
    nearly all of it.  Same decisions as the sequential oracle (SceneDensify.cpp:3395-3449), in both pixel orders of the C-ABI's raster
    rule (order 0)."""
    maps, order = chain_maps()
    upload(ctx, maps)
    want = O.fuse_depthmaps(maps, order, 20000, n_min_views_fuse=3)
    got = ctx.fuse(order, 20000, n_min_views_fuse=3)
    h, w = maps[0]["depth"].shape
    assert want["n_points"] == h * (w // 2)                 # every other pixel of A, nothing from B and C
    assert got["n_points"] == want["n_points"] and got["n_depths"] == want["n_depths"]
    assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["n_views"], want["n_views"])
    for i, d in enumerate(want["depths"]):
        assert np.array_equal(ctx.get_depthmap(i)[0], d)


@pytest.mark.parametrize("kw,fk", [
    (dict(w=640, h=480, f=600.0, n_views=9, noise=0.008, outliers=0.15, holes=0.05, seed=12), dict(n_min_views_fuse=3)),
    (dict(w=512, h=384, f=480.0, n_views=12, noise=0.003, outliers=0.05, holes=0.3, seed=13, far=2, far_factor=2.2), dict(depthweight=2.0, normalweight=1.5))])
def test_fuse_bit_exact_large_dirty_maps(ctx, kw, fk):
    """bigger and dirtier than the scenes above (15 % outliers, 30 % holes, a coarse neighbour, 11 neighbours per image; nMinViewsFuse 3;
    the depth / normal weights of FuseDepthMaps): many pixels do NOT become points here, so the settle iteration starts far from its
    fixed point.  Cloud, point order and invalidated depths equal the sequential oracle's."""
    maps, order = make_maps(**kw)
    upload(ctx, maps)
    cap = kw["w"] * kw["h"] * kw["n_views"]
    want = O.fuse_depthmaps(maps, order, cap, **fk)
    got = ctx.fuse(order, cap, **fk)
    assert got["n_points"] == want["n_points"] > 100000 and got["n_depths"] == want["n_depths"]
    assert np.array_equal(got["xyz"], want["xyz"]) and np.array_equal(got["n_views"], want["n_views"])
    assert np.array_equal(got["normal"], want["normal"]) and np.array_equal(got["bgr"], want["bgr"])
    for i, d in enumerate(want["depths"]):
        assert np.array_equal(ctx.get_depthmap(i)[0], d)


def test_fuse_count_then_fuse_is_the_single_pass(ctx):
    """hcmvs_fuse_cloud without buffers counts (and invalidates); the fusion that follows, with buffers of exactly the counted size, yields
    the cloud of a single pass -- a fusion repeated on the maps a fusion has left makes the same decisions (what it invalidated is simply
    absent the second time).  The property the driver's two-pass fusion and the incremental post-filter chain rest on."""
    for kw, mode in ((dict(noise=0.003, outliers=0.05, holes=0.1), 0), (dict(w=144, h=112, f=130.0, n_views=7, noise=0.002, outliers=0.03), 0),
                     (dict(noise=0.003, outliers=0.05, holes=0.1), 1)):
        maps, order = make_maps(**kw)
        ctx.set_fuse_order(mode)
        upload(ctx, maps)
        single = ctx.fuse_cloud(order, 200000, 800000)
        after_single = [ctx.get_depthmap(i)[0] for i in range(len(maps))]
        upload(ctx, maps)
        n_points, n_depths, n_entries = ctx.fuse_count(order)
        assert (n_points, n_depths, n_entries) == (single["n_points"], single["n_depths"], len(single["view_ids"]))
        for i in range(len(maps)):
            assert np.array_equal(ctx.get_depthmap(i)[0], after_single[i])       # the counting pass has invalidated what the single pass invalidates
        again = ctx.fuse_cloud(order, n_points, n_entries)                       # exactly the counted sizes
        assert again["n_points"] == n_points
        for k in ("xyz", "normal", "bgr", "n_views", "view_ids", "view_weights"):
            assert np.array_equal(again[k], single[k]), k
        for i in range(len(maps)):
            assert np.array_equal(ctx.get_depthmap(i)[0], after_single[i])       # ... and the second pass invalidates nothing more
        assert ctx.fuse_count(order)[0] == n_points                              # nor a third
    ctx.set_fuse_order(0)
