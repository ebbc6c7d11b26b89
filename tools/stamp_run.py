"""diagnostic (not a test): per-phase cycle shares of the sweep workers, using the stamps build (batch of B images)"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
binding = importlib.import_module("hc-mvs_amd.binding")
binding.LIB_PATH = binding.LIB_PATH.replace("libhcmvs_hip.so", "libhcmvs_hip_stamps.so")
synth = importlib.import_module("hc-mvs_amd.synth")
W, H, F, V = 1920, 1080, 1600.0, 8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
I = int(sys.argv[2]) if len(sys.argv) > 2 else 4   # sweeps (bench.py: 8)
AUTHORS = len(sys.argv) > 3 and sys.argv[3] == "authors"  # 10 source views, half window 7, outer iteration 1 with the cross pattern 5 / 4
if AUTHORS: V = 10
dev = torch.device("cuda:0")
ctx = binding.Context(0)
items, keep = [], []
for b in range(B):
    views = synth.make_views(W, H, F, V, seed=2 + b); pts = synth.sparse_points(views, 2000)
    g = [torch.from_numpy(v["gray"]).to(dev) for v in views]
    for i, v in enumerate(views): ctx.set_view_device(100 * b + i, W, H, g[i].data_ptr(), v["K"], v["R"], v["C"])
    ctx.shapes[100 * b] = (H, W)
    d0, n0, dmin, dmax = ctx.splat_init(100 * b, pts)
    init = torch.cat([torch.from_numpy(d0).reshape(-1), torch.from_numpy(n0).reshape(-1), torch.zeros(H * W)]).to(dev)
    work = torch.empty_like(init)
    HW = H * W
    items.append(dict(ref_id=100 * b, src_ids=[100 * b + i for i in range(1, V + 1)], d_min=dmin, d_max=dmax, d_depth=work.data_ptr(),
                      d_normal=work.data_ptr() + 4 * HW, d_conf=work.data_ptr() + 16 * HW, seed_offset=b))
    keep.append((g, init, work))
p = binding.default_params(adapthalfwin=6, n_estimation_iters=I)
if AUTHORS:
    kw = dict(adapthalfwin=7, n_estimation_iters=I, n_external_iters=4, propagate_halfwin=5, propagate_step=4, photometric_flow=0.26, seed=4321)
    for g, init, work in keep: work.copy_(init)
    torch.cuda.synchronize()
    ctx.estimate_batch_device(items, binding.default_params(it_external=0, **kw))  # outer iteration 0 makes the input maps
    torch.cuda.synchronize()
    for k, (g, init, work) in enumerate(keep): init.copy_(work)
    p = binding.default_params(it_external=1, **kw)
L = binding.lib(); L.hcmvs_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 32)()
for rep in range(2):
    for g, init, work in keep: work.copy_(init)
    torch.cuda.synchronize()
    L.hcmvs_debug_stamps(ctx._h, out, 1)
    ctx.estimate_batch_device(items, p)
    torch.cuda.synchronize()
st = ctx.stats()
L.hcmvs_debug_stamps(ctx._h, out, 1)
names = ["0 wait for the row above", "1 fill_patch (after 12)", "2 slots + interp + park", "3 hypothesis generation", "4 smooth_pass",
         "5 (prop tail)", "6 publish hook", "7 score_chunk", "8 exchange", "9 wait for the row above to BEGIN (ramp)", "10 ticket", "11 accept scan", "12 loop top + loads issue", "13 loop back-edge (end of process_pixel -> loop top)", "14 wait for the stretch to my left", "15 store + ring + publish"]
tot = sum(out[i] for i in range(16))
npx = (W - 14) * (H - 14) * I * B
print("B", B, "ms_sweep_avg %.2f" % st.ms_sweep_avg, "s_memtime ticks/pixel (wave0) %.0f" % (tot / npx))
for i in range(16):
    print("%-32s %6.1f%%  %8.0f ticks/px" % (names[i], 100.0 * out[i] / tot, out[i] / npx))

# executions of the blocks of the pixel state machine (BLOCK() in pm_kernels.hip), per pixel-sweep of wave 0's rows
cn = ["pick", "gen_prop", "gen_rand", "gen_refine", "chunk", "score_chunk", "acc_prop", "acc_prop_cand", "acc_rand", "acc_refine", "smooth_chunks"]
print("blocks per pixel-sweep:", ", ".join("%s=%.4f" % (n, out[16 + i] / npx) for i, n in enumerate(cn)))
print("evals per pixel-sweep: sequential %.4f issued %.4f" % (st.evals / (npx + (W - 14) * (H - 14) * B) * (1 + 1.0 / I), st.evals_issued / npx))
print("stats: evals %d issued %d (incl. the init-score pass: %d)" % (st.evals, st.evals_issued, (W - 14) * (H - 14) * B))
