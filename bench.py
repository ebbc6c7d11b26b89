#!/usr/bin/env python3
"""bench.py -- PatchMatch depth-map estimation throughput on MI355X (BASELINE.json metric).

One "step" = one complete EstimateDepthMap of one reference image (median + init-score pass + 8
propagate/refine sweeps + end pass; SceneDensify.cpp:758-1072) on BASELINE.json configs[1]:
1 reference x 8 source views, 1920x1080, 7x7 taps (adapthalfwin 6), 8 sweeps, synthetic pinhole scene.
Inputs (images, initial maps) are resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank estimates its own reference image (independent units, no collective on the estimation
path) and the packed {depth, normal, conf} maps are all-gathered over RCCL after each step, which is the
exchange FuseDepthMaps needs (SceneDensify.cpp:3381-3449).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, FOCAL, N_SRC, SWEEPS, AHW = 1920, 1080, 1600.0, 8, 8, 6
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def cpu_baseline(n_threads):
    """The CPU oracle (restatement of the reference algorithm, reference arithmetic) on a bounded sample of
    the same workload: 960x540, 8 source views, 7x7, 8 sweeps, row-pipelined over n_threads host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    synth = importlib.import_module("hc-mvs_amd.synth")
    w, h = 960, 540
    views = synth.make_views(w, h, FOCAL / 2, N_SRC, seed=2)
    pts = synth.sparse_points(views, 500)
    L = O.lib()
    ref = O.make_view(views[0])
    d0 = np.zeros((h, w), np.float32); n0 = np.zeros((h, w, 3), np.float32)
    dmin = ctypes.c_float(); dmax = ctypes.c_float()
    L.hcor_splat_init(ctypes.byref(ref), O.fptr(pts), len(pts), O.fptr(d0), O.fptr(n0), ctypes.byref(dmin),
                      ctypes.byref(dmax))
    p = O.default_params(adapthalfwin=AHW, n_estimation_iters=SWEEPS, arith_mode=O.ARITH_REFERENCE,
                         order=O.ORDER_ROWS, n_threads=n_threads)
    t0 = time.time()
    O.estimate(views, p, dmin.value, dmax.value, d0, n0)
    dt = time.time() - t0
    return {"value": round(w * h / dt / 1e6, 4), "unit": "Mpix/s", "cores": n_threads, "kind": "port",
            "sample": "960x540 synthetic scene, 8 source views, 7x7 taps, 8 sweeps, one full estimate (%.1f s)" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"

    binding = importlib.import_module("hc-mvs_amd.binding")  # imports torch first: one HIP runtime per process
    synth = importlib.import_module("hc-mvs_amd.synth")
    D = importlib.import_module("hc-mvs_amd.distributed")

    # synthetic scene: every rank gets its own reference image + source views (weak scaling)
    views = synth.make_views(W, H, FOCAL, N_SRC, seed=2 + rank)
    pts = synth.sparse_points(views, 2000, seed=5 + rank)
    ctx = binding.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    d_gray = []
    for i, v in enumerate(views):
        t = torch.from_numpy(v["gray"]).to(dev)
        d_gray.append(t)
        ctx.set_view_device(i, W, H, t.data_ptr(), v["K"], v["R"], v["C"])
    # initial maps (SceneDensify.cpp:783-808 splat of the sparse points), resident on the device
    ctx.shapes[0] = (H, W)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    init = torch.cat([torch.from_numpy(d0).reshape(-1), torch.from_numpy(n0).reshape(-1),
                      torch.zeros(H * W)]).to(dev)
    work = torch.empty_like(init)
    HW = H * W
    esz = work.element_size()
    p_depth, p_normal, p_conf = work.data_ptr(), work.data_ptr() + HW * esz, work.data_ptr() + 4 * HW * esz
    params = binding.default_params(adapthalfwin=AHW, n_estimation_iters=SWEEPS, it_external=0, n_external_iters=1,
                                    seed=1234)
    src_ids = list(range(1, N_SRC + 1))

    def step():
        work.copy_(init)
        ctx.estimate_device(0, src_ids, params, dmin, dmax, p_depth, p_normal, p_conf)
        if world > 1:  # the exchange FuseDepthMaps needs: every rank receives every map (20 B/px)
            D.allgather_maps(work.view(1, -1))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = ctx.stats()  # of the last step: HIP events recorded on the stream the kernels ran on

    if rank == 0:
        P = (W - 14) * (H - 14)
        gra = ctx.gradient_map(0)[7:H - 7, 7:W - 7]
        taps_px = np.where(gra > 100, 36, (AHW + 1) ** 2).astype(np.int64)
        tap_evals_sweeps = int(st.tap_evals) - int(taps_px.sum())  # pass A scores every pixel once
        # algorithmic bytes of ONE sweep launch (SURVEY.md 8d tap-gather convention): every bilinear sample
        # counts its 4 texels (16 B) per source view, plus the 24 B of per-pixel state read and 20 B written
        bytes_sweep = tap_evals_sweeps / SWEEPS * N_SRC * 16.0 + P * (24.0 + 20.0)
        achieved = bytes_sweep / (st.ms_sweep_avg * 1e-3) / 1e9
        out = {
            "metric": "PatchMatch Mpix/s (1080p, 8 views, 7x7, 8 iter)",
            "value": round(world * W * H * args.steps / dt / 1e6, 4),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1 ref x 8 src views per GPU, 1920x1080, 7x7 taps (adapthalfwin 6), 8 sweeps, "
                                   "it_external 0, full EstimateDepthMap (median + init score + sweeps + end pass)",
                       "units_per_step": "one reference image per GPU (%d px)" % (W * H),
                       "exchange": "RCCL all-gather of 20 B/px maps per step" if world > 1 else "none",
                       "evals_per_pixel_sweep": round((st.evals / P - 1) / SWEEPS, 3)},
            "per_gpu": round(W * H * args.steps / dt / 1e6, 4),
            "kernel_ms": {"score_pass": round(st.ms_score, 3), "sweep_avg": round(st.ms_sweep_avg, 3),
                          "sweeps_total": round(st.ms_sweeps, 3), "end": round(st.ms_end, 3),
                          "estimate_total": round(st.ms_total, 3)},
            "roofline": {"kernel": "sweep_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None,
                         "algorithmic_bytes_per_launch": int(bytes_sweep),
                         "avg_launch_ms": round(st.ms_sweep_avg, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 16))
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
