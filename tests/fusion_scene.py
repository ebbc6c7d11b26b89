"""helper: a ring of views with ground-truth (optionally perturbed) maps for the filter / fuse tests"""
import importlib

import numpy as np

synth = importlib.import_module("hc-mvs_amd.synth")


def make_maps(w=96, h=80, f=90.0, n_views=5, seed=3, noise=0.0, outliers=0.0, holes=0.0, far=None, far_factor=2.8):
    """far: index of a view that is moved back to far_factor times the scene distance (same focal length): its pixels are
    far_factor times coarser, so several pixels of the other views land on each of its pixels"""
    px = 10.0 / f
    scene = synth.Scene(seed, min_wavelength=3.5 * px, max_wavelength=150 * px)
    views = synth.make_views(w, h, f, n_views - 1, seed=seed, baseline=(0.04, 0.09), scene=scene)
    if far is not None:
        v = views[far]
        C = np.array([v["C"][0], v["C"][1], -(far_factor - 1.0) * scene.depth0])
        R = synth.look_at(C, np.array([0.0, 0.0, scene.depth0]))
        gray, depth, normal = scene.render(v["K"], R, C, w, h)
        views[far] = dict(K=v["K"], R=R, C=C, gray=gray, depth=depth, normal=normal, width=w, height=h)
    rng = np.random.RandomState(seed + 77)
    maps = []
    for i, v in enumerate(views):
        d = v["depth"].copy()
        d[~np.isfinite(d)] = 0
        if noise:
            d *= (1 + noise * rng.normal(size=d.shape)).astype(np.float32)
        if outliers:
            m = rng.uniform(size=d.shape) < outliers
            d[m] *= rng.uniform(0.6, 1.5, size=m.sum()).astype(np.float32)
        if holes:
            d[rng.uniform(size=d.shape) < holes] = 0
        d[:7] = 0; d[-7:] = 0; d[:, :7] = 0; d[:, -7:] = 0   # the estimator leaves a 7 px border empty
        conf = np.where(d > 0, rng.uniform(0.5, 0.95, size=d.shape), 0).astype(np.float32)
        g8 = np.clip(np.rint(v["gray"] * 255), 0, 255).astype(np.uint8)
        bgr = np.stack([g8, np.roll(g8, 1, 1), 255 - g8], -1).copy()
        nb = [j for j in range(n_views) if j != i]
        # decreasing importance: closest cameras first
        nb.sort(key=lambda j: np.linalg.norm(views[j]["C"] - v["C"]))
        maps.append(dict(K=v["K"], R=v["R"], C=v["C"], gray=v["gray"], depth=d.astype(np.float32),
                         normal=v["normal"].copy(), conf=conf, bgr=bgr, d_min=float(v["depth"].min() * 0.8),
                         d_max=float(v["depth"].max() * 1.2), neighbors=nb, gt=v["depth"]))
    order = sorted(range(n_views), key=lambda i: (-len(maps[i]["neighbors"]), i))
    return maps, order
