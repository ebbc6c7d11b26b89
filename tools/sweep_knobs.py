"""diagnostic (not a test): batched sweep time of the bench workload (B x 1080p, 8 views, 7x7) under different tuning knobs
and diagnostic builds, one process:  python tools/sweep_knobs.py B SWEEPS "lib:lag[:affinity[:nw]]" ...
  lib = '' (the product library) | occ4 | occ2 | ... (libhcmvs_hip_<lib>.so)"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")
W, H, F, V = 1920, 1080, 1600.0, int(os.environ.get("HCMVS_KNOB_VIEWS", "8"))  # source views per reference image
if os.environ.get("HCMVS_KNOB_SIZE"):  # e.g. 3840x2160
    W, H = (int(v) for v in os.environ["HCMVS_KNOB_SIZE"].split("x")); F = 1600.0 * W / 1920
B, I = int(sys.argv[1]), int(sys.argv[2])
configs = sys.argv[3:] or [":1"]
dev = torch.device("cuda:0")
scenes = []
for s in range(4):
    views = synth.make_views(W, H, F, V, seed=2 + s)
    scenes.append((views, synth.sparse_points(views, 2000, seed=5 + s), torch.from_numpy(np.stack([v["gray"] for v in views])).to(dev)))
base = binding.LIB_PATH
HW = H * W
allwork = torch.empty(B, 5 * HW, dtype=torch.float32, device=dev)
for cfg in configs:
    parts = cfg.split(":")
    libname, lag = parts[0], parts[1] if len(parts) > 1 else "1"
    os.environ["HCMVS_SWEEP_LAG"] = lag
    os.environ["HCMVS_XCD_AFFINITY"] = parts[2] if len(parts) > 2 and parts[2] else "1"
    if len(parts) > 3 and parts[3]: os.environ["HCMVS_WAVES_PER_ROW"] = parts[3]
    else: os.environ.pop("HCMVS_WAVES_PER_ROW", None)
    binding._lib = None
    binding.LIB_PATH = base.replace("libhcmvs_hip.so", "libhcmvs_hip_%s.so" % libname) if libname else base
    ctx = binding.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    items, inits = [], []
    for b in range(B):
        views, pts, slab = scenes[b % 4]
        for i, v in enumerate(views): ctx.set_view_device(100 * b + i, W, H, slab[i].data_ptr(), v["K"], v["R"], v["C"])
        ctx.shapes[100 * b] = (H, W)
        d0, n0, dmin, dmax = ctx.splat_init(100 * b, pts)
        inits.append(torch.cat([torch.from_numpy(d0).reshape(-1), torch.from_numpy(n0).reshape(-1), torch.zeros(HW)]).to(dev))
        w = allwork[b]
        items.append(dict(ref_id=100 * b, src_ids=[100 * b + i for i in range(1, V + 1)], d_min=dmin, d_max=dmax, d_depth=w.data_ptr(),
                          d_normal=w.data_ptr() + 4 * HW, d_conf=w.data_ptr() + 16 * HW, seed_offset=b))
    p = binding.default_params(adapthalfwin=6, n_estimation_iters=I)
    best = None
    for rep in range(2):
        for b in range(B): allwork[b].copy_(inits[b])
        torch.cuda.synchronize()
        ctx.estimate_batch_device(items, p)
        torch.cuda.synchronize()
        st = ctx.stats()
        if best is None or st.ms_sweep_avg < best[0]: best = (st.ms_sweep_avg, st.ms_total, st.evals, float(allwork[0, :HW].sum()))
    print("cfg %-16s B=%d sweeps=%d: sweep launch avg %.2f ms, estimate %.1f ms -> %.2f Mpix/s at 8 sweeps (evals %d, checksum %.6g)" % (
        cfg, B, I, best[0], best[1], B * W * H / (best[1] + (8 - I) * best[0]) / 1e3, best[2], best[3]), flush=True)
    ctx.close()
