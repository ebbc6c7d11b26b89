"""ad-hoc: FilterDepthMap throughput (reference SceneDensify.cpp:3006-3259) on a ring of 1080p views"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fusion_scene import make_maps
binding = importlib.import_module("hc-mvs_amd.binding")
W, H, N = 1920, 1080, 9
maps, order = make_maps(w=W, h=H, f=1600.0, n_views=N, noise=0.002, outliers=0.03, holes=0.05)
ctx = binding.Context(0)
for i, m in enumerate(maps):
    ctx.upload_view(i, m["gray"], m["K"], m["R"], m["C"], bgr=m["bgr"])
    ctx.set_depthmap(i, m["depth"], m["normal"], m["conf"], m["d_min"], m["d_max"])
for adjust in (True, False):
    for rep in range(2):
        t = time.time()
        d, c, nproc, ndisc = ctx.filter(0, maps[0]["neighbors"][:8], adjust=adjust)
        dt = time.time() - t
    print("filter 1080p, 8 neighbours, adjust=%s: %.2f ms (%.1f Mpix/s incl. copying the two maps back), processed %d discarded %d" % (
        adjust, dt * 1e3, W * H / dt / 1e6, nproc, ndisc))
