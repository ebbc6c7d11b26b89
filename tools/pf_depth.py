"""diagnostic (not a test): what a fusion pass costs on the maps the post-filters see -- the UNFILTERED score maps between outer
iterations (every pixel holds an estimate) -- against the final, end-filtered maps: pending pixels, points, steps and work-list sizes
of the settle iteration, time per image (the HCMVS_FUSE_DEBUG lines: one host synchronisation per image).
  python tools/pf_depth.py [n_views=9]"""
import importlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np, torch
    binding = importlib.import_module("hc-mvs_amd.binding")
    if os.environ.get("HCMVS_LIB"):   # a diagnostic build of the library
        binding.LIB_PATH = binding.LIB_PATH.replace("libhcmvs_hip.so", os.environ["HCMVS_LIB"])
    synth = importlib.import_module("hc-mvs_amd.synth")
    W, H, F = 1920, 1080, 1600.0
    n = int(sys.argv[1])
    dev = torch.device("cuda:0")
    views = synth.make_views(W, H, F, n - 1, seed=2)
    pts = synth.sparse_points(views, 2000, seed=5)
    ctx = binding.Context(0)
    HW = H * W
    for i, v in enumerate(views):
        g8 = np.clip(np.rint(v["gray"] * 255), 0, 255).astype(np.uint8)
        ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"], bgr=np.stack([g8, g8, g8], -1).copy())
    work = torch.zeros(n, 5 * HW, dtype=torch.float32, device=dev)
    items, rng = [], []
    for i in range(n):
        d0, n0, dmin, dmax = ctx.splat_init(i, pts)
        work[i, :HW] = torch.from_numpy(d0).reshape(-1).to(dev); work[i, HW:4 * HW] = torch.from_numpy(n0).reshape(-1).to(dev)
        base = work[i].data_ptr()
        items.append(dict(ref_id=i, src_ids=[j for j in range(n) if j != i], d_min=dmin, d_max=dmax, d_depth=base, d_normal=base + 4 * HW, d_conf=base + 16 * HW, seed_offset=i))
        rng.append((dmin, dmax))
    for final in (0, 1):   # 0: maps as they stand after outer iteration 1 of 4 (what the post-filters fuse); 1: final maps
        for it in range(2):
            p = binding.default_params(adapthalfwin=7, n_estimation_iters=3, seed=7, it_external=it, n_external_iters=2 if final else 4,
                                       propagate_halfwin=5, propagate_step=4, photometric_flow=0.26)
            if it == 0:
                for i in range(n):
                    d0, n0, _, _ = ctx.splat_init(i, pts)
                    work[i, :HW] = torch.from_numpy(d0).reshape(-1).to(dev); work[i, HW:4 * HW] = torch.from_numpy(n0).reshape(-1).to(dev); work[i, 4 * HW:] = 0
            torch.cuda.synchronize()
            ctx.estimate_batch_device(items, p)
            ctx.synchronize()
        for i in range(n):
            base = work[i].data_ptr()
            ctx.set_depthmap_device(i, base, base + 4 * HW, base + 16 * HW, rng[i][0], rng[i][1])
            ctx.set_neighbors(i, [j for j in range(n) if j != i])
        print("MAPS", "final" if final else "between outer iterations", "valid fraction %.3f" % float((work[:, :HW] > 0).float().mean()), flush=True)
        got = ctx.fuse(list(range(n)), HW * n // 2)
        print("POINTS", got["n_points"], flush=True)
    ctx.close()
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "9"
r = subprocess.run([sys.executable, os.path.abspath(__file__), n, "child"], env=dict(os.environ, HCMVS_FUSE_DEBUG="1"), capture_output=True, text=True)
for ln in (r.stdout + r.stderr).splitlines():
    if ln.startswith(("MAPS", "POINTS", "fuse:")):
        print(ln)
