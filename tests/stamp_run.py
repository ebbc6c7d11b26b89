"""diagnostic (not a test): per-phase cycle shares of the sweep workers, using the stamps build"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
binding = importlib.import_module("hc-mvs_amd.binding")
binding.LIB_PATH = binding.LIB_PATH.replace("libhcmvs_hip.so", "libhcmvs_hip_stamps.so")
synth = importlib.import_module("hc-mvs_amd.synth")
W, H, F, V, I = 1920, 1080, 1600.0, 8, 4
views = synth.make_views(W, H, F, V, seed=2); pts = synth.sparse_points(views, 2000)
ctx = binding.Context(0)
for i, v in enumerate(views): ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
d0, n0, dmin, dmax = ctx.splat_init(0, pts)
p = binding.default_params(adapthalfwin=6, n_estimation_iters=I)
L = binding.lib(); L.hcmvs_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 16)()
ctx.estimate(0, list(range(1, V + 1)), p, dmin, dmax, d0, n0)
L.hcmvs_debug_stamps(ctx._h, out, 1)
ctx.estimate(0, list(range(1, V + 1)), p, dmin, dmax, d0, n0)
st = ctx.stats()
L.hcmvs_debug_stamps(ctx._h, out, 1)
names = ["0 loop top (wait/poll/upload)", "1 fill_patch", "2 slots+interp", "3 prop score", "4 prop exchange", "5 prop scan",
         "6 hook1", "7 refine hyp+score", "8 refine exchange", "9 refine scan+store", "10 row end/ticket", "11"]
tot = sum(out[i] for i in range(12))
npx = (W - 14) * (H - 14) * I
print("NW", os.environ.get("HCMVS_WAVES_PER_ROW", "4"), "ms_sweep_avg %.2f" % st.ms_sweep_avg, "cycles/pixel (wave0) %.0f" % (tot / npx))
for i in range(11):
    print("%-32s %6.1f%%  %8.0f cyc/px" % (names[i], 100.0 * out[i] / tot, out[i] / npx))
