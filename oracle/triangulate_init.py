"""oracle/triangulate_init.py -- TEST INFRASTRUCTURE (part of the CPU oracle; never imported by the product).

numpy/scipy restatement of the reference's triangulation initialisation (TriangulatePointsDelaunay +
TriangulatePoints2DepthMap, frame_main/libs/MVS/DepthMap.cpp:1796-1936; rasteriser TImage::RasterizeTriangle,
frame_main/libs/Common/Types.inl:2474-2606).  The reference triangulates with CGAL (absent here); a Delaunay
triangulation is unique for points in general position, so scipy.spatial.Delaunay (Qhull) serves as the independent
triangulator.  Parity unpinned: the reference ships no fixture for this step.
"""
import numpy as np
from scipy.spatial import Delaunay


def _i2c(K, x, y, z):  # Camera.h:306-312
    return np.array([(x - K[0, 2]) * z / K[0, 0], (y - K[1, 2]) * z / K[1, 1], z], np.float64)


def triangulate_init(w, h, K, R, C, xyz, avg_depth=0.0, add_corners=True):
    """returns depth (h,w) f32, normal (h,w,3) f32, d_min, d_max (range of the projected points, widened by 0.9/1.1 as
    DepthMapsData::InitDepthMap does, SceneDensify.cpp:523-525)"""
    K = np.asarray(K, np.float64); R = np.asarray(R, np.float64); C = np.asarray(C, np.float64)
    P = np.hstack([K @ R, -(K @ R @ C)[:, None]])
    X = np.asarray(xyz, np.float64)
    p = (P[:, :3] @ X.T).T + P[:, 3]
    p = p.astype(np.float32)                                   # ProjectPointP3 -> Point3f
    keep = p[:, 2] > 0
    p = p[keep]
    q = np.stack([p[:, 0] / p[:, 2], p[:, 1] / p[:, 2], p[:, 2]], -1).astype(np.float64)   # (x/z, y/z, z), float division
    lo, hi = np.float32(q[:, 2].min()), np.float32(q[:, 2].max())
    if not avg_depth > 0:
        avg_depth = np.float32(q[:, 2].sum() / len(xyz))
    if add_corners:
        inside = (q[:, 0] >= 0) & (q[:, 1] >= 0) & (q[:, 0] <= w) & (q[:, 1] <= h)
        q = q[inside]
    _, first = np.unique(q[:, :2], axis=0, return_index=True)  # CGAL keeps the first point inserted at a position
    q = q[np.sort(first)]
    corners = np.array([[0, 0, avg_depth], [w, 0, avg_depth], [w, h, avg_depth], [0, h, avg_depth]], np.float64)
    pts = np.vstack([corners, q]) if add_corners else q
    tri = Delaunay(pts[:, :2])
    faces = tri.simplices
    if add_corners:  # DepthMap.cpp:1810-1876
        for c in range(4):
            A = pts[c]
            ray = _i2c(K, A[0], A[1], A[2]); ray /= np.linalg.norm(ray)
            cand = []
            for t in np.nonzero((faces == c).any(1))[0]:
                k = int(np.nonzero(faces[t] == c)[0][0])
                g = tri.neighbors[t][k]                         # the face opposite the corner
                if g < 0:
                    continue
                G = faces[g]
                if (G < 4).any():
                    continue
                c0, c1, c2 = (_i2c(K, *pts[v]) for v in G)
                n = np.cross(c1 - c0, c2 - c0); n /= np.linalg.norm(n)
                z = ray[2] * (n @ c0) / (n @ ray)
                if not z > 0:
                    continue
                b = (pts[G[0], :2] + pts[G[1], :2] + pts[G[2], :2]) / np.float32(3)
                dist = np.linalg.norm(b - A[:2])
                cand.append((np.float32(1) / np.float32(dist), np.clip(np.float32(z), lo, hi)))
            if len(cand) < 3:
                continue
            cand.sort(key=lambda s: -s[0])
            wts = np.array([s[0] for s in cand[:3]], np.float32); wts = wts * (np.float32(1) / wts.sum(dtype=np.float32))
            pts[c, 2] = np.float32(sum(np.float32(s[1]) * wk for s, wk in zip(cand[:3], wts)))
    depth = np.zeros((h, w), np.float32); normal = np.zeros((h, w, 3), np.float32)
    fx, fy, cx, cy = (np.float32(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
    r16 = lambda v: int(np.floor(16.0 * float(np.float32(v)) + 0.5))
    for f in faces:
        a, b, c = pts[f[0]], pts[f[1]], pts[f[2]]
        if (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]) < 0:
            f = f[[0, 2, 1]]; a, b, c = pts[f[0]], pts[f[1]], pts[f[2]]
        cc = [np.array([(np.float64(np.float32(v[0])) - K[0, 2]) * np.float64(np.float32(v[2])) / K[0, 0],
                        (np.float64(np.float32(v[1])) - K[1, 2]) * np.float64(np.float32(v[2])) / K[1, 1], np.float32(v[2])], np.float32) for v in (a, b, c)]
        n = np.cross(cc[2] - cc[0], cc[1] - cc[0]).astype(np.float32)
        nl = np.float32(np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]))
        if not nl > 0:
            continue
        n = n / nl
        npl = n * (np.float32(1) / np.float32(n[0] * cc[0][0] + n[1] * cc[0][1] + n[2] * cc[0][2]))
        v1, v2, v3 = c, b, a                                    # reversed, as the reference passes it
        X1, X2, X3, Y1, Y2, Y3 = r16(v1[0]), r16(v2[0]), r16(v3[0]), r16(v1[1]), r16(v2[1]), r16(v3[1])
        DX12, DX23, DX31, DY12, DY23, DY31 = X1 - X2, X2 - X3, X3 - X1, Y1 - Y2, Y2 - Y3, Y3 - Y1
        minx, maxx = (min(X1, X2, X3) + 15) >> 4, (max(X1, X2, X3) + 15) >> 4
        miny, maxy = (min(Y1, Y2, Y3) + 15) >> 4, (max(Y1, Y2, Y3) + 15) >> 4
        minx &= ~7; miny &= ~7
        C1, C2, C3 = DY12 * X1 - DX12 * Y1, DY23 * X2 - DX23 * Y2, DY31 * X3 - DX31 * Y3
        C1 += DY12 < 0 or (DY12 == 0 and DX12 > 0); C2 += DY23 < 0 or (DY23 == 0 and DX23 > 0); C3 += DY31 < 0 or (DY31 == 0 and DX31 > 0)
        y0, y1 = max(miny, 0), min(miny + ((maxy - miny + 7) // 8) * 8, h)
        x0, x1 = max(minx, 0), min(minx + ((maxx - minx + 7) // 8) * 8, w)
        if y1 <= y0 or x1 <= x0:
            continue
        ys, xs = np.mgrid[y0:y1, x0:x1]
        XS, YS = xs.astype(np.int64) << 4, ys.astype(np.int64) << 4
        m = (C1 + DX12 * YS - DY12 * XS > 0) & (C2 + DX23 * YS - DY23 * XS > 0) & (C3 + DX31 * YS - DY31 * XS > 0)
        X0x = (xs.astype(np.float32) - cx) / fx; X0y = (ys.astype(np.float32) - cy) / fy
        z = np.float32(1) / (npl[0] * X0x + npl[1] * X0y + npl[2] * np.float32(1))
        m &= z > 0
        depth[ys[m], xs[m]] = z[m]
        normal[ys[m], xs[m]] = n
    return depth, normal, float(lo * np.float32(0.9)), float(hi * np.float32(1.1))
