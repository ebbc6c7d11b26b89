"""Known-answer / property tests of the oracle's FilterDepthMap and FuseDepthMaps restatements."""
import numpy as np

import oracle_lib as O
from fusion_scene import make_maps


def test_fuse_perfect_maps_lie_on_the_surface():
    maps, order = make_maps(noise=0.0)
    r = O.fuse_depthmaps(maps, order, 60000)
    valid_depths = sum(int((m["depth"] > 0).sum()) for m in maps)
    assert r["n_depths"] <= valid_depths and r["n_points"] > 0.15 * valid_depths
    assert np.all(r["n_views"] >= 2) and r["n_views"].max() <= 5
    # every fused point reprojects onto the ground-truth depth of view 0 where it is visible
    m = maps[0]
    Xc = (r["xyz"].astype(np.float64) - m["C"]) @ m["R"].T
    px = np.rint(m["K"][0, 2] + m["K"][0, 0] * Xc[:, 0] / Xc[:, 2]).astype(int)
    py = np.rint(m["K"][1, 2] + m["K"][1, 1] * Xc[:, 1] / Xc[:, 2]).astype(int)
    ins = (px >= 8) & (py >= 8) & (px < 88) & (py < 72) & (Xc[:, 2] > 0)
    rel = np.abs(Xc[ins, 2] - m["gt"][py[ins], px[ins]]) / m["gt"][py[ins], px[ins]]
    assert np.median(rel) < 2e-3 and (rel < 0.02).mean() > 0.9  # discrete reprojection near the sphere edge aside
    assert np.allclose(np.linalg.norm(r["normal"], axis=1), 1, atol=1e-4)


def test_fuse_is_greedy_and_order_dependent():
    maps, order = make_maps(noise=0.002)
    a = O.fuse_depthmaps(maps, order, 60000)
    b = O.fuse_depthmaps(maps, order[::-1], 60000)
    assert a["n_points"] != b["n_points"] or not np.array_equal(a["xyz"], b["xyz"])
    assert abs(a["n_points"] - b["n_points"]) < 0.1 * a["n_points"]
    # a point needs nMinViewsFuse views: with a threshold nobody can meet nothing is fused (SceneDensify.cpp:3426)
    c = O.fuse_depthmaps(maps, order, 60000, n_min_views_fuse=6)
    assert c["n_points"] == 0
    # an isolated image (no neighbours) cannot reach 2 views either
    for m in maps:
        m["neighbors"] = []
    assert O.fuse_depthmaps(maps, order, 60000)["n_points"] == 0


def test_fuse_invalidates_occluding_estimates():
    maps, order = make_maps(noise=0.0, outliers=0.05)
    r = O.fuse_depthmaps(maps, order, 60000)
    zeroed = sum(int(((m["depth"] > 0) & (d == 0)).sum()) for m, d in zip(maps, r["depths"]))
    assert zeroed > 0  # estimates in front of accepted points were discarded (SceneDensify.cpp:3447-3449)
    clean = O.fuse_depthmaps(make_maps(noise=0.0)[0], order, 60000)
    assert sum(int(((m["depth"] > 0) & (d == 0)).sum()) for m, d in zip(make_maps(noise=0.0)[0], clean["depths"])) < zeroed


def test_filter_keeps_consistent_and_drops_outliers():
    maps, _ = make_maps(noise=0.001, outliers=0.08)
    nb = maps[0]["neighbors"]
    ok, d, c, nproc, ndisc = O.filter_depthmap(maps, 0, nb, adjust=True)
    assert ok == 1 and nproc == int((maps[0]["depth"] > 0).sum()) and 0 < ndisc < nproc
    gt = maps[0]["gt"]; was_out = (maps[0]["depth"] > 0) & (np.abs(maps[0]["depth"] - gt) / gt > 0.15)
    kept_out = was_out & (d > 0)
    assert kept_out.sum() < 0.35 * was_out.sum()  # most gross outliers lose the vote
    good = (d > 0) & ~was_out
    assert (np.abs(d - gt)[good] / gt[good] < 0.01).mean() > 0.8  # averaging accepts neighbours within 12 % (SceneDensify.cpp:3127)
    # the strict variant keeps depths untouched where it keeps them
    ok, d2, c2, _, _ = O.filter_depthmap(maps, 0, nb, adjust=False)
    assert np.all((d2 == 0) | (d2 == maps[0]["depth"])) and (d2 > 0).sum() > 0
    # too few neighbours: the reference returns false (SceneDensify.cpp:3016-3019)
    assert O.filter_depthmap(maps, 0, nb[:1], adjust=True, n_min_views=2)[0] == 0
