"""ad-hoc: FuseDepthMaps throughput on a ring of N 1080p views (ground-truth maps + noise/outliers/holes), GPU vs the
CPU oracle on a reduced sample"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fusion_scene import make_maps
binding = importlib.import_module("hc-mvs_amd.binding")
W, H, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = time.time()
maps, order = make_maps(w=W, h=H, f=1600.0 * W / 1920, n_views=N, noise=0.002, outliers=0.03, holes=0.05)
print("scene %.1f s" % (time.time() - t), flush=True)
ctx = binding.Context(0)
for rep in range(2):
    for i, m in enumerate(maps):
        ctx.upload_view(i, m["gray"], m["K"], m["R"], m["C"], bgr=m["bgr"])
        ctx.set_depthmap(i, m["depth"], m["normal"], m["conf"], m["d_min"], m["d_max"])
        ctx.set_neighbors(i, m["neighbors"][:8])
    ctx.synchronize() if hasattr(ctx, "synchronize") else None
    t = time.time()
    got = ctx.fuse(order, W * H * N // 2)
    dt = time.time() - t
    print("GPU fuse %dx%d x%d: %.3f s, %d points from %d depths -> %.2f Mpoints/s, %.1f Mdepths/s" % (
        W, H, N, dt, got["n_points"], got["n_depths"], got["n_points"] / dt / 1e6, got["n_depths"] / dt / 1e6), flush=True)
if len(sys.argv) > 4:
    import oracle_lib as O
    for m in maps: m["neighbors"] = m["neighbors"][:8]
    t = time.time()
    want = O.fuse_depthmaps(maps, order, W * H * N // 2)
    dt = time.time() - t
    print("CPU oracle fuse: %.3f s, %d points -> %.2f Mpoints/s; equal count %s" % (dt, want["n_points"], want["n_points"] / dt / 1e6, want["n_points"] == got["n_points"]))
