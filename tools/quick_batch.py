"""ad-hoc: one batched estimate of B reference images (hcmvs_estimate_batch_device) -> aggregate throughput"""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")
W, H, F, V, I, B = 1920, 1080, 1600.0, 8, 8, int(sys.argv[1])
dev = torch.device("cuda:0")
ctx = binding.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
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
torch.cuda.synchronize()
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    t = time.time()
    for g, init, work in keep: work.copy_(init)
    ctx.estimate_batch_device(items, p)
    torch.cuda.synchronize()
    dt = time.time() - t
    st = ctx.stats()
    print("B=%d NW=%s rep %d: %.3f s -> %.2f Mpix/s aggregate (sweep launch avg %.2f ms, score %.1f ms)" % (B, os.environ.get("HCMVS_WAVES_PER_ROW", "2"), rep, dt, B * W * H / dt / 1e6, st.ms_sweep_avg, st.ms_score), flush=True)
