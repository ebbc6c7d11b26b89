"""Known answers of the oracle's restatement of the fork's post-filters (oracle/hcmvs_fuse.c hcor_postfilter; reference
RemoveSmallSegments as rewritten + GapInterpolation, frame_main/libs/MVS/SceneDensify.cpp:2048-2275, 2280-3001, applied at
:3939-3958).  No fixture exists in the reference (parity unpinned): the vectors follow from the cited lines."""
import numpy as np

import oracle_lib as O
from fusion_scene import make_maps


def _plane_maps():
    """two views of a fronto-parallel plane at depth 10: every pixel of view 0 fuses with view 1"""
    w, h, f = 64, 48, 60.0
    K = np.array([[f, 0, (w - 1) / 2], [0, f, (h - 1) / 2], [0, 0, 1.0]])
    maps = []
    for i in range(2):
        d = np.full((h, w), 10.0, np.float32)
        n = np.zeros((h, w, 3), np.float32); n[..., 2] = -1
        maps.append(dict(K=K, R=np.eye(3), C=np.array([0.1 * i, 0, 0]), depth=d, normal=n, conf=np.full((h, w), 0.8, np.float32), bgr=None,
                         d_min=1.0, d_max=100.0, neighbors=[1 - i]))
    return maps, w, h


def test_short_gaps_are_interpolated_and_unfused_pixels_keep_their_depth():
    maps, w, h = _plane_maps()
    m0 = maps[0]
    m0["depth"][20, 10:14] = 0                 # a 4-pixel hole in a row: ends agree -> filled
    m0["depth"][30, 20:40] = 0                 # a 20-pixel hole: longer than nIpolGapSize = 7 ...
    m0["depth"][5, 30] = 13.0                  # an estimate no other view confirms: not fused = a one-pixel gap of the fused map
    m0["conf"][20, 9] = 0.6
    gra = np.full((h, w), 50, np.uint8)        # ... but the gradient map is flat (ratio 0 <= 0.1): filled as well
    depths, n, c, filled = O.postfilter(maps, 0, gra, [0, 1], mode=O.ARITH_REFERENCE)
    d = depths[0]
    # the image is shifted by one pixel against view 1 (baseline 0.1, f = 60, depth 10 -> 0.6 px): everything else fuses
    assert np.allclose(d[20, 10:14], 10.0, atol=1e-5) and np.allclose(c[20, 10:14], 0.6)
    assert np.allclose(d[30, 20:40], 10.0, atol=1e-5)
    assert abs(d[5, 30] - 10.0) < 1e-5         # ... which the interpolation fills: the inconsistent estimate is replaced (SD.cpp:2989-3000)
    assert np.allclose(n[20, 10:14], [0, 0, -1], atol=1e-6)
    assert filled >= 24
    # a sloped row: the fill is the linear interpolation between the ends (depthFirst + k * (depth - depthFirst) / (count + 1))
    maps, w, h = _plane_maps()
    for mm in maps:
        mm["depth"][:] = (10.0 + 0.002 * np.arange(w))[None, :].astype(np.float32)
    want = maps[0]["depth"][25].copy()
    maps[0]["depth"][25, 30:35] = 0
    depths, n, c, filled = O.postfilter(maps, 0, gra, [0, 1], mode=O.ARITH_REFERENCE)
    a, b = want[29], want[35]
    lin = a + (b - a) / 6 * np.arange(1, 6)
    assert np.allclose(depths[0][25, 30:35], lin, rtol=1e-6)


def test_long_gaps_need_similar_gradient_or_depth():
    maps, w, h = _plane_maps()
    for mm in maps:
        mm["depth"][:, 32:] = 12.0             # a depth step at column 32
    maps[0]["depth"][5:, 22:42] = 0            # a hole across the step, 20 wide, open to the lower border (columns have no second end)
    gra = np.full((h, w), 50, np.uint8)
    gra[5:, 42] = 80                           # the gradient map at the right ends differs by 60 %, the depths by 20 % > 2.5 %: stays open
    depths, _, _, _ = O.postfilter(maps, 0, gra, [0, 1], mode=O.ARITH_REFERENCE)
    assert (depths[0][5:, 22:42] == 0).all()
    gra[5:, 42] = 54                           # 8 % -> the rows are filled despite the depth step
    depths, _, _, _ = O.postfilter(maps, 0, gra, [0, 1], mode=O.ARITH_REFERENCE)
    row = depths[0][10, 22:42]
    assert (row > 0).all() and row[0] < row[-1] and abs(row[0] - (10 + 2 / 21)) < 1e-4
    maps[0]["depth"][:, 0:5] = 0               # a hole touching the left border has no first end: left open (u > count fails)
    depths, _, _, _ = O.postfilter(maps, 0, gra, [0, 1], mode=O.ARITH_REFERENCE)
    assert (depths[0][:, 0:5] == 0).all()


def test_fusion_side_effects_and_modes_agree():
    maps, order = make_maps(w=96, h=80, f=90.0, n_views=4, noise=0.002, outliers=0.05, holes=0.1)
    gra = (np.random.RandomState(1).uniform(0, 255, maps[0]["depth"].shape)).astype(np.uint8)
    dr, nr, cr, fr = O.postfilter(maps, 1, gra, order, mode=O.ARITH_REFERENCE)
    dd, nd, cd, fd = O.postfilter(maps, 1, gra, order, mode=O.ARITH_DEVICE)
    assert fr == fd > 50
    f = O.fuse_depthmaps(maps, order, 100000)
    for i in (0, 2, 3):                        # the other images are changed exactly as a plain fusion changes them
        assert np.array_equal(dr[i], f["depths"][i])
    assert (dr[1] > 0).sum() > (f["depths"][1] > 0).sum()          # holes of image 1 were filled
    assert np.abs(dr[1] - dd[1]).max() < 1e-5 and np.abs(nr - nd).max() < 1e-5 and np.array_equal(cr, cd)


def gap_scene(pairs):
    """Two views of a plane with a depth step (10 -> 12 at column 32, seen alike by both views, so both sides fuse) and, in
    view 0, a hole 20 pixels wide across the step that is open to the lower border (columns have no second end): too long for
    nIpolGapSize = 7 and with ends 20 % apart, so only the gradient-map rule `texture_ratio <= 0.1` (SD.cpp:2383-2390, columns
    :2713-2720) can fill a row of it.  Returns (maps, the rows whose hole ends carry the gradient values `pairs`, the end
    columns, the 8-bit image whose gradient map has exactly those values there and 0 at the ends of every other row)."""
    maps, w, h = _plane_maps()
    for mm in maps:
        mm["depth"][:, 32:] = 12.0
    rows = [8 + 6 * k for k in range(len(pairs))]
    x0, x1 = 21, 42
    maps[0]["depth"][6:, x0 + 1:x1] = 0
    img = np.full((h, w), 50, np.uint8)
    for y, (g0, g1) in zip(rows, pairs):
        # Sobel x on rows y-1..y+1 that are alike: gx = 4 (v[x+1] - v[x-1]), gy = 0, gra = |gx| / 2; an extra +1 at (x+1, y) alone
        # adds 2 to gx without touching gy, which makes the odd values
        for x, g in ((x0, g0), (x1, g1)):
            img[y - 1:y + 2, x + 1:] += g // 2
            if g % 2:
                img[y, x + 1] += 1
    return maps, rows, (x0, x1), img


def test_gradient_ratio_is_compared_with_the_double_literal():
    """SD.cpp:2390 / :2720 test the FLOAT ratio against the DOUBLE literal 0.1: a ratio that rounds to 0.1f (gradient pairs
    (10, 11), (20, 22), (90, 99)) is > 0.1 and must NOT fill; (20, 21) fills."""
    pairs = [(10, 11), (20, 22), (90, 99)]
    for pr, filled in ((pairs, False), ([(20, 21)], True)):
        maps, rows, (x0, x1), img = gap_scene(pr)
        gra = O.gradient_map(img.astype(np.float32) / 255.0)     # the GPU test uploads img; the device computes this map itself
        for y, (g0, g1) in zip(rows, pr):
            assert (int(gra[y, x0]), int(gra[y, x1])) == (g0, g1)
            assert (np.float32(g1 - g0) / np.float32(g0) == np.float32(0.1)) == (not filled)
        for mode in (O.ARITH_REFERENCE, O.ARITH_DEVICE):
            depths, _, _, _ = O.postfilter(maps, 0, gra, [0, 1], mode=mode)
            for y in rows:
                if filled:
                    assert (depths[0][y, x0 + 1:x1] > 0).all()
                else:
                    assert (depths[0][y, x0 + 1:x1] == 0).all(), "a ratio of exactly 0.1f must not fill (mode %d, row %d)" % (mode, y)
