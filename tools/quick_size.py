"""ad-hoc: one estimate at an arbitrary size (e.g. 4K / 12 MP of BASELINE.json configs[3], [4]): runs, converges, and does
not depend on the schedule (one vs two waves per row)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")
W, H, V, I = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
views = synth.make_views(W, H, 1600.0 * W / 1920, V, seed=2); pts = synth.sparse_points(views, 4000)
res = []
for nw in ("1", "2"):
    os.environ["HCMVS_WAVES_PER_ROW"] = nw
    ctx = binding.Context(0)
    for i, v in enumerate(views): ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
    d0, n0, dmin, dmax = ctx.triangulate_init(0, pts)
    p = binding.default_params(adapthalfwin=6, n_estimation_iters=I)
    t = time.time()
    d, n, c = ctx.estimate(0, list(range(1, V + 1)), p, dmin, dmax, d0, n0)
    dt = time.time() - t
    st = ctx.stats()
    gt = views[0]["depth"]; m = d > 0
    print("%dx%d V=%d %d sweeps, %s wave(s)/row: %.2f s (%.2f Mpix/s incl. PCIe), valid %.3f, within 1%% %.3f, evals/px/sweep %.2f" % (
        W, H, V, I, nw, dt, W * H / dt / 1e6, m.mean(), (np.abs(d - gt)[m] / gt[m] < 0.01).mean(), (st.evals / ((W - 14) * (H - 14)) - 1) / I), flush=True)
    res.append((d, n, c)); ctx.close()
print("schedule invariant:", all(np.array_equal(a, b) for a, b in zip(res[0], res[1])))
