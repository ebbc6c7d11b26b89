"""Known-answer checks of oracle/select_views.py (restatement of Scene::SelectNeighborViews / FilterNeighborViews /
DepthMapsData::InitViews, frame_main/libs/MVS/Scene.cpp:531-678, SceneDensify.cpp:307-397).  The reference holds no fixture
for this step (parity unpinned): the vectors below are derived by hand from the cited formulas."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import select_views as SV  # noqa: E402


def cam(C, f=100.0, w=160, h=120, target=(0.0, 0.0, 10.0)):
    C = np.asarray(C, np.float64)
    z = np.asarray(target, np.float64) - C; z /= np.linalg.norm(z)
    x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return dict(K=np.array([[f, 0, (w - 1) / 2], [0, f, (h - 1) / 2], [0, 0, 1.0]]), R=np.stack([x, y, z]), C=C)


def grid_points(n=6, z=10.0, spread=3.0):
    xs = np.linspace(-spread, spread, n)
    return [np.array([x, y * 0.7, z], np.float32) for x in xs for y in xs]


def test_scores_follow_the_formulas():
    # B sits where every ray to the points makes ~10 degrees with A's (the optimum: wAngle = 1), C at ~2 degrees
    # (wAngle = (2/10)^1.5), D twice as far from the points (footprint ratio 2 > 1.6 -> wScale = (1.6/2)^2)
    b10 = 10.0 * np.tan(np.deg2rad(10.0))
    b2 = 10.0 * np.tan(np.deg2rad(2.0))
    cams = [cam((0, 0, 0)), cam((b10, 0, 0)), cam((b2, 0, 0)), cam((0, 0, -10.0))]
    sizes = [(160, 120)] * 4
    pts = [np.array([0.0, 0.0, 10.0], np.float32)] * 5      # five copies of the point on the optical axis
    verts = [(p, [0, 1, 2, 3]) for p in pts]
    points, nb, ok = SV.select_neighbor_views(cams, sizes, verts, 0)
    assert points == [0, 1, 2, 3, 4] and ok
    by = {n["id"]: n for n in nb}
    area = 1.0 / 256                                          # all projections fall into one cell of the 16 x 16 grid
    assert by[1]["angle"] == pytest.approx(np.deg2rad(10.0), rel=1e-5) and by[1]["scale"] == pytest.approx(1.0 / np.cos(np.deg2rad(10.0)), rel=1e-5)
    # B is slightly farther from the point than A (depth 10/cos(10 deg)): footprint ratio r = 1/cos > 1 -> wScale = 1
    assert by[1]["score"] == pytest.approx(5 * 1.0 * area, rel=1e-5)
    r2 = 1.0 / np.cos(np.deg2rad(2.0))
    assert by[2]["scale"] == pytest.approx(r2, rel=1e-5)
    assert by[2]["score"] == pytest.approx(5 * (2.0 / 10.0) ** 1.5 * area, rel=1e-4)
    assert by[3]["scale"] == pytest.approx(2.0, rel=1e-6) and by[3]["angle"] == pytest.approx(0.0, abs=1e-3)
    assert by[3]["score"] == pytest.approx(0.0, abs=1e-6)    # zero angle -> zero weight
    assert [n["id"] for n in nb][:2] == [1, 2]               # sorted by decreasing score


def test_filter_and_init_views():
    nb = [dict(id=1, points=9, scale=1.0, angle=np.deg2rad(12), area=0.5, score=4.0),
          dict(id=2, points=9, scale=3.3, angle=np.deg2rad(12), area=0.5, score=3.0),    # scale out of [0.2, 3.2)
          dict(id=3, points=9, scale=1.0, angle=np.deg2rad(2), area=0.5, score=2.5),     # angle below 3 degrees
          dict(id=4, points=9, scale=1.3, angle=np.deg2rad(30), area=0.005, score=2.0),  # area below 0.01
          dict(id=5, points=9, scale=1.3, angle=np.deg2rad(30), area=0.2, score=1.0),
          dict(id=6, points=9, scale=0.9, angle=np.deg2rad(64.9), area=0.2, score=0.1)]  # score below 3 % of the best
    kept = SV.filter_neighbor_views(nb)
    assert [n["id"] for n in kept] == [1, 5, 6]
    assert SV.init_views(kept, 5) == [(1, 1.0), (5, 1.3)]    # 6 fails the score ratio; 5 is resampled (|1.3 - 1| >= 0.15)
    assert SV.init_views(kept, 1) == [(1, 1.0)]              # number-views caps the list
    assert [n["id"] for n in SV.filter_neighbor_views(nb * 5, n_max_views=4)] == [1, 5, 6, 1]


def test_covered_area_and_visibility_rules():
    cams = [cam((0, 0, 0)), cam((1.5, 0, 0)), cam((-1.5, 0.3, 0)), None]
    sizes = [(160, 120)] * 4
    P = grid_points()
    verts = [(p, [0, 1] if i % 2 else [0, 1, 2]) for i, p in enumerate(P)]
    verts.append((np.array([50.0, 0, 10.0], np.float32), [0, 1, 2]))   # projects outside image 0: no grid cell
    verts.append((np.array([0.0, 0, 12.0], np.float32), [1, 2]))       # not seen by image 0: ignored
    points, nb, ok = SV.select_neighbor_views(cams, sizes, verts, 0)
    assert ok and len(points) == 37 and 37 not in points
    by = {n["id"]: n for n in nb}
    assert by[1]["points"] == 37 and by[2]["points"] == 19
    # covered area = occupied cells / 256 of the points shared with B that project inside both images
    f, cx, cy = 100.0, 79.5, 59.5
    cells = {(int((f * p[0] / p[2] + cx) / 160 * 16), int((f * p[1] / p[2] + cy) / 120 * 16)) for p in P}
    assert by[1]["area"] == pytest.approx(len(cells) / 256.0)
    sel = SV.select(cams, sizes, verts, 0, number_views=5)
    assert [s[0] for s in sel["srcs"]] == [1, 2] and all(s[1] == 1.0 for s in sel["srcs"])
    # fewer than 3 shared points with an image -> it is no neighbour; too few neighbours -> failure
    assert SV.select(cams[:2] + [None, None], sizes, [(p, [0, 1]) for p in P[:2]], 0) is None
