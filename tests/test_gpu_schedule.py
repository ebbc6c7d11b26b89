"""The schedule of the fork's post-filters over the images of an outer iteration, on the device, against the
scene-level oracle harness (tests/scene_oracle.py) in the SAME schedule -- bit for bit:

  batch        estimate every image, then filter image after image (hcmvs_postfilter_sequence): the device path's default
  interleaved  estimate(k) -> post-filter(k) -> estimate(k + 1): the reference's single-thread event order
               (SceneDensify.cpp:3889-3965), `densify_scene(interleave=True)` / `DensifyPointCloud --n-postfilter-interleave 1`

through the C-ABI binding (densify_scene) and through the stand-alone driver (files in, files out)."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
import scene_files as SF
import scene_oracle as S

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import select_views as SV  # noqa: E402

pytestmark = pytest.mark.gpu
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")
mvsio = importlib.import_module("hc-mvs_amd.mvsio")
D = importlib.import_module("hc-mvs_amd.distributed")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "hc-mvs_amd", "DensifyPointCloud")


@pytest.mark.parametrize("interleave", [False, True], ids=["batch", "interleaved"])
def test_scene_schedule_matches_the_oracle_in_that_schedule(interleave):
    """five images, three outer iterations (cross pattern from the second on), post-filters after outer iterations 1 and 2, fusion;
    set_fuse_order(1) is active the whole time: the post-filters' fusion must stay in raster order (SceneDensify.cpp:2130-2131) --
    only the final FuseDepthMaps follows the option, and that is compared against the oracle in the hashed order too"""
    import torch
    views, srcs, neighbors, order, init = S.ring_scene(n=5, w=128, h=96, f=120.0, n_points=80)
    kw = dict(adapthalfwin=6, n_estimation_iters=2, propagate_halfwin=5, propagate_step=4)
    want = S.densify(views, srcs, neighbors, order, init, n_external_iters=3, postfilter=True, interleave=interleave, seed=900, fuse=False, **kw)
    ctx = binding.Context(0)
    try:
        ctx.set_fuse_order(1)
        p = binding.default_params(seed=900, **kw)
        cloud = D.densify_scene(ctx, views, srcs, neighbors, order, init, p, device=torch.device("cuda", 0), n_external_iters=3, postfilter=True,
                                interleave=interleave)
        # the maps as they were BEFORE the final fusion touched them are gone (fusion invalidates depths in place), so the oracle fuses too
        maps = [dict(K=views[i]["K"], R=views[i]["R"], C=views[i]["C"], depth=want["maps"][i][0], normal=want["maps"][i][1], conf=want["maps"][i][2],
                     bgr=views[i]["bgr"], d_min=init[i][2], d_max=init[i][3], neighbors=neighbors[i]) for i in order]
        fused = O.fuse_depthmaps(maps, order, 128 * 96 * 5, pixel_order=1)
        assert cloud["n_points"] == fused["n_points"] > 3000 and cloud["n_depths"] == fused["n_depths"]
        for i in order:
            d, n, c = [t.cpu().numpy() for t in cloud["maps"][i]]
            assert np.array_equal(d, fused["depths"][i]), "depth map %d" % i
            assert np.array_equal(n, want["maps"][i][1]) and np.array_equal(c, want["maps"][i][2]), "normal / confidence map %d" % i
        # hashed order: the same points as sets (the device writes them in raster order, the oracle in visiting order)
        key = lambda x: x[np.lexsort(x.T[::-1])]
        assert np.array_equal(key(cloud["xyz"]), key(fused["xyz"]))
    finally:
        ctx.close()


def _driver_scene(tmp, n=5, w=192, h=144):
    f = 230.0 * w / 192
    px = 10.0 / f
    scene = synth.Scene(7, min_wavelength=3.5 * px, max_wavelength=150 * px)
    K = np.array([[f, 0, (w - 1) / 2.0], [0, f, (h - 1) / 2.0], [0, 0, 1]], np.float64)
    target = np.array([0.0, 0.0, scene.depth0])
    poses = []
    for i in range(n):
        ang = 2 * np.pi * i / n
        Cc = np.array([0.7 * np.cos(ang), 0.5 * np.sin(ang), 0.03 * i])
        poses.append((synth.look_at(Cc, target), Cc))
    views = SF.render_views(scene, K, poses, w, h, threads=4)
    verts = SF.sparse_vertices(views, 160, seed=2)
    return views, verts, SF.write_scene(tmp, views, verts)


@pytest.mark.parametrize("interleave", [0, 1], ids=["batch", "interleaved"])
def test_driver_postfilter_schedules_match_the_oracle(tmp_path, interleave):
    """DensifyPointCloud with three outer iterations and the post-filters on (--n-nOptimize 2, the default), in the batch schedule and
    with --n-postfilter-interleave 1: final DR depth maps and the fused .ply against the scene-level oracle in the same schedule
    (same view selection, splat initialisation, seeds), bit for bit."""
    assert os.path.exists(EXE), "build the driver first: make -C hc-mvs_amd/csrc"
    tmp = str(tmp_path)
    views, verts, scene_path = _driver_scene(tmp)
    n = len(views)
    out = os.path.join(tmp, "dense.mvs")
    seed = 777
    r = subprocess.run([EXE, "-i", scene_path, "-o", out, "--resolution-level", "0", "--min-resolution", "64", "--number-views", "3", "--n-EstimationIters", "2",
                        "--n-EstimationIters-external", "3", "--n-adapthalfwin", "6", "--n-propagatehalfwin", "5", "--n-propagatestep", "4",
                        "--n-photometric_flow", "0", "--min-views-trust-point", "1", "--n-postfilter-interleave", str(interleave), "--seed", str(seed), "-v", "3"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ("image after image" in r.stdout) == bool(interleave) and ("Depth-maps filtered after outer iteration" in r.stdout) == (not interleave)
    cams = [dict(K=v["K"], R=v["R"], C=v["C"]) for v in views]
    sizes = [(v["width"], v["height"]) for v in views]
    vlist = [(x["X"], [j for j, _ in x["views"]]) for x in verts]
    g8 = [np.clip(np.rint(v["gray"] * 255), 0, 255).astype(np.uint8) for v in views]
    oviews, srcs, neighbors, init = {}, {}, {}, {}
    for i in range(n):
        sel = SV.select(cams, sizes, vlist, i, number_views=3)
        assert sel is not None
        srcs[i] = [s[0] for s in sel["srcs"]]
        assert all(abs(s[1] - 1) < 0.15 for s in sel["srcs"])                      # no resampled neighbour in this scene
        neighbors[i] = [nb["id"] for nb in sel["neighbors"]]
        oviews[i] = dict(K=views[i]["K"], R=views[i]["R"], C=views[i]["C"], gray=SF.driver_gray(g8[i]), bgr=np.stack([g8[i]] * 3, -1).copy())
        pts = np.ascontiguousarray(np.stack([verts[k]["X"] for k in sel["points"]]), np.float32)
        init[i] = S.splat(oviews[i], pts)
    order = sorted(range(n), key=lambda i: -len(neighbors[i]))                     # stable: best connected first (SceneDensify.cpp:3302)
    want = S.densify(oviews, srcs, neighbors, order, init, n_external_iters=3, postfilter=True, interleave=bool(interleave), seed=seed, n_threads=16,
                     adapthalfwin=6, n_estimation_iters=2, propagate_halfwin=5, propagate_step=4, photometric_flow=0.0)
    for i in range(n):
        dm = mvsio.read_dmap(os.path.join(tmp, "depth%04d.dmap" % i))
        assert list(dm["ids"]) == [i] + srcs[i]
        assert np.array_equal(dm["depth"], want["maps"][i][0]), "depth map %d differs from the oracle (%s schedule)" % (i, "interleaved" if interleave else "batch")
        assert np.array_equal(dm["normal"], want["maps"][i][1]) and np.array_equal(dm["conf"], want["maps"][i][2])
        m = dm["depth"] > 0
        gt = views[i]["depth"]
        assert m.mean() > 0.6 and (np.abs(dm["depth"] - gt)[m] / gt[m] < 0.01).mean() > 0.85
    ply = mvsio.read_ply(out[:-4] + ".ply")
    xyz = np.stack([ply["x"], ply["y"], ply["z"]], -1)
    assert len(xyz) == want["cloud"]["n_points"] > 10000 and np.array_equal(xyz, want["cloud"]["xyz"])
    m = re.search(r"(\d+) depth-maps, (\d+) depths, (\d+) points", r.stdout)
    assert m and int(m.group(3)) == want["cloud"]["n_points"]
