"""ad-hoc timing helper (not a test): python tests/quick_time.py W H F V SWEEPS"""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
binding = importlib.import_module("hc-mvs_amd.binding")
if os.environ.get("HCMVS_LIB"): binding.LIB_PATH = binding.LIB_PATH.replace("libhcmvs_hip.so", os.environ["HCMVS_LIB"])
synth = importlib.import_module("hc-mvs_amd.synth")
W, H, F, V, I = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
t = time.time()
views = synth.make_views(W, H, F, V, seed=2)
pts = synth.sparse_points(views, 2000)
print("scene %.1fs" % (time.time() - t), flush=True)
ctx = binding.Context(0)
for i, v in enumerate(views):
    ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
d0, n0, dmin, dmax = ctx.splat_init(0, pts)
p = binding.default_params(adapthalfwin=6, n_estimation_iters=I)
for rep in range(2):
    t = time.time()
    d, n, c = ctx.estimate(0, list(range(1, V + 1)), p, dmin, dmax, d0, n0)
    st = ctx.stats()
    P = (W - 14) * (H - 14)
    print("rep %d wall %.3fs  ms_total %.1f score %.1f sweeps %.1f (avg %.2f) end %.2f evals %d (%.2f/px/sweep, issued %.2fx) Mpix/s %.2f" %
          (rep, time.time() - t, st.ms_total, st.ms_score, st.ms_sweeps, st.ms_sweep_avg, st.ms_end, st.evals,
           (st.evals / P - 1) / max(I, 1), st.evals_issued / st.evals, W * H / st.ms_total / 1e3), flush=True)
gt = views[0]["depth"]; valid = d > 0
rel = np.abs(d - gt) / gt
print("valid %.3f rel<1%% %.3f" % (valid.mean(), (rel[valid] < 0.01).mean()))
