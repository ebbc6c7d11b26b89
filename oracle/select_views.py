"""oracle/select_views.py -- TEST INFRASTRUCTURE (part of the CPU oracle; never imported by the product).

numpy restatement of how the reference chooses the source views of a reference image:
  Scene::SelectNeighborViews   frame_main/libs/MVS/Scene.cpp:531-661  (Footprint :531-539, scoring :575-603, covered area
                               :606-645 with ComputeCoveredArea<float,2,16,false>, libs/Common/Util.inl:711-730)
  Scene::FilterNeighborViews   Scene.cpp:665-678, called with the bounds of DepthMapsData::SelectViews
                               (SceneDensify.cpp:307-327: fMinArea 0.01, scale [0.2, 3.2), angle [3, 65) degrees, nMaxViews 12;
                               defaults DepthMap.cpp:72-86)
  DepthMapsData::InitViews     SceneDensify.cpp:336-397: neighbours in score order while #images <= number-views and
                               score >= best * fViewMinScoreRatio * 0.1; a neighbour whose average scale differs from 1 by
                               >= 0.15 is resampled by that scale (DepthMap.h:233-238), which the returned `scale` reports.
Parity unpinned: the reference ships no fixture for this step; the restatement follows the cited lines, in float where the
reference computes in float.
"""
import numpy as np

F32 = np.float32


def _w2c(cam, X):
    return cam["R"] @ (np.asarray(X, np.float64) - cam["C"])


def _project(cam, X):
    """Camera::ProjectPointP (Camera.h:284-288) for a Point3f: P = K [R | -R C] applied in double, result as float"""
    K, R, C = cam["K"], cam["R"], cam["C"]
    q = K @ (R @ (np.asarray(X, np.float64) - C))
    return F32(q[0] / q[2]), F32(q[1] / q[2])


def select_neighbor_views(cams, sizes, verts, ID, n_min_views=2, n_min_point_views=2, optim_angle_deg=10.0):
    """cams: list of dict(K, R, C) float64 (None for an uncalibrated image); sizes: list of (w, h); verts: list of
    (X float32[3], [image ids sorted]).  Returns (points, neighbors, ok): points = indices of the vertices seen by ID and
    by >= n_min_point_views images; neighbors = list of dict(id, points, scale, angle, area, score) sorted by decreasing
    score (stable); ok = the reference's return value (Scene.cpp:655-660)."""
    n_cal = sum(c is not None for c in cams)
    n_min_point_views = min(n_min_point_views, n_cal)
    f_optim = F32(np.deg2rad(F32(optim_angle_deg)))
    A = cams[ID]
    n_img = len(cams)
    score = np.zeros(n_img, F32); avg_scale = np.zeros(n_img, F32); avg_angle = np.zeros(n_img, F32)
    cnt = np.zeros(n_img, np.int64)
    points = []
    for idx, (X, views) in enumerate(verts):
        if ID not in views:
            continue
        if len(views) >= n_min_point_views:
            points.append(idx)
        X = np.asarray(X, F32)
        V1 = (A["C"] - X.astype(np.float64)).astype(F32)
        fp1 = F32(A["K"][0, 0] / _w2c(A, X)[2])                       # Footprint: focal / depth
        for v in views:
            if v == ID:
                continue
            B = cams[v]
            V2 = (B["C"] - X.astype(np.float64)).astype(F32)
            ca = F32(np.dot(V1, V2)) / F32(np.sqrt(F32(np.dot(V1, V1)) * F32(np.dot(V2, V2))))   # Util.inl:417-420
            ang = F32(np.arccos(np.clip(ca, F32(-1), F32(1))))
            w_angle = min(F32(np.power(ang / f_optim, F32(1.5))), F32(1))
            fp2 = F32(B["K"][0, 0] / _w2c(B, X)[2])
            r = fp1 / fp2
            if r > F32(1.6):
                w_scale = (F32(1.6) / r) ** 2
            elif r >= F32(1):
                w_scale = F32(1)
            else:
                w_scale = r * r
            score[v] += F32(w_angle * w_scale); avg_scale[v] += r; avg_angle[v] += ang; cnt[v] += 1
    neighbors = []
    wA, hA = sizes[ID]
    for IDB in range(n_img):
        if cams[IDB] is None or cnt[IDB] < 3:
            continue
        wB, hB = sizes[IDB]
        grid = np.zeros((16, 16), bool)
        n_proj = 0
        for idx in points:
            X, views = verts[idx]
            if IDB not in views:
                continue
            if _w2c(A, X)[2] <= 0 or _w2c(cams[IDB], X)[2] <= 0:
                continue
            ua, va = _project(A, X); ub, vb = _project(cams[IDB], X)
            if not (ua >= 0 and va >= 0 and ua < wA and va < hA and ub >= 0 and vb >= 0 and ub < wB and vb < hB):
                continue
            grid[int(np.floor(ua / F32(wA) * F32(16))), int(np.floor(va / F32(hA) * F32(16)))] = True
            n_proj += 1
        if n_proj == 0:
            continue
        area = F32(grid.sum()) / F32(256)
        neighbors.append(dict(id=IDB, points=int(cnt[IDB]), scale=float(avg_scale[IDB] / F32(cnt[IDB])),
                              angle=float(avg_angle[IDB] / F32(cnt[IDB])), area=float(area), score=float(score[IDB] * area)))
    neighbors.sort(key=lambda n: -n["score"])                          # stable, decreasing score (Types.h:2432)
    ok = len(points) > 3 and len(neighbors) >= min(n_min_views, n_cal - 1)
    return points, neighbors, ok


def filter_neighbor_views(neighbors, min_area=0.01, min_scale=0.2, max_scale=3.2, min_angle_deg=3.0, max_angle_deg=65.0, n_max_views=12):
    lo, hi = float(F32(np.deg2rad(F32(min_angle_deg)))), float(F32(np.deg2rad(F32(max_angle_deg))))
    kept = [n for n in neighbors if not (n["area"] < min_area) and min_scale <= n["scale"] < max_scale and lo <= n["angle"] < hi]
    return kept[:n_max_views]


def init_views(neighbors, number_views, min_score_ratio=0.3, min_score=0.0):
    """SceneDensify.cpp:362-375: returns the chosen source views as (id, scale) pairs; scale != 1 means the view is
    resampled (|scale - 1| >= 0.15, DepthMap.h:233-238)"""
    if not neighbors:
        return []
    f_min = max(neighbors[0]["score"] * (min_score_ratio * 0.1), min_score)
    out = []
    for n in neighbors:
        if (number_views and len(out) + 1 > number_views) or n["score"] < f_min:
            break
        out.append((n["id"], n["scale"] if abs(n["scale"] - 1.0) >= 0.15 else 1.0))
    return out


def select(cams, sizes, verts, ID, number_views=5):
    """SelectViews + InitViews for image ID: dict(points, neighbors (filtered), srcs [(id, scale)]) or None"""
    points, nb, ok = select_neighbor_views(cams, sizes, verts, ID)
    if not ok:
        return None
    nb = filter_neighbor_views(nb)
    if not nb:
        return None
    srcs = init_views(nb, number_views)
    if not srcs:
        return None
    return dict(points=points, neighbors=nb, srcs=srcs)
