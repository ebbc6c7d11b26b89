#!/usr/bin/env python3
"""bench.py -- PatchMatch depth-map estimation throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one BATCH of synthetic input: --batch B (default 32) independent units
of BASELINE.json configs[1] -- 1 reference x 8 source views, 1920x1080, 7x7 taps (adapthalfwin 6), 8 sweeps --
each a complete EstimateDepthMap (median + init-score pass + 8 propagate/refine sweeps + end pass;
SceneDensify.cpp:758-1072) of its own synthetic pinhole scene, issued as one hcmvs_estimate_batch_device call
(the reference likewise overlaps images, SceneDensify.cpp:3699).  Inputs (images, initial maps) are resident in HBM
before the timed region starts; every unit has its own buffers and its own RNG stream, the synthetic scene content
repeats every 4 units (rendering 32 distinct 1080p scenes on the host would take minutes).  The single-unit latency
(B = 1) is reported next to it as "single_unit".

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank estimates its own reference image (independent units, no collective on the estimation
path) and the packed {depth, normal, conf} maps are all-gathered over RCCL after each step, which is the
exchange FuseDepthMaps needs (SceneDensify.cpp:3381-3449).  Prints ONE JSON line on rank 0.

Beside the contract fields the line carries: `roofline` = the bound that limits the sweep kernel, the VECTOR ALU: `achieved` = wave64
VALU instructions per second (SQ_INSTS_VALU of one launch / its duration), `peak` = 1024 SIMDs x 2.4 GHz / 2 cycles, `frac` = the
issue fraction, `useful_flop_frac` = the flops SURVEY.md 8d counts (1.6 kflop per evaluation and view) over the 157.3 TFLOP/s vector
peak, `traffic` = HBM bytes per launch; the counters are measured NOW by rocprofv3 --pmc child runs of this same command, one pass per
counter set (`counter_source` says so, or names the committed profile used when rocprofv3 is not usable).  `roofline_convention` keeps
the SURVEY.md 8d tap-gather figure (algorithmic bytes over the HBM peak; it passes 1.0 because the gather is served by L2 -- printed,
not capped).  A batch of 16 or more images runs all its sweeps in ONE kernel launch: "per launch" then means per 8 sweeps
(`roofline.sweeps_per_launch`, `kernel_ms.sweep_launches`).  Also `roofline_single_unit`, `fuse` (FuseDepthMaps points/s on estimated maps,
with `fuse.roofline`: the SURVEY.md 8d bytes over its wall time, bound "latency", the share of the cloud's copy to the host), `authors_config`
(the reference authors' own settings -- 10 views, 8x8 taps, cross pattern 5 / 4 -- as a batch of 16: the TWO + PACK kernel instance, time per
sweep and its VALU fractions from a rocprofv3 child run of `bench.py --authors-only`), `cpu_baseline` (the oracle on the host cores).
The default run takes about three and a half minutes.
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
N_SIMD, CLOCK_HZ = 1024, 2.4e9        # 256 CUs x 4 SIMDs at the 2.4 GHz the kernel holds (GRBM_GUI_ACTIVE, MI355X_MICROARCH.md)
VALU_PEAK_INSTS = N_SIMD * CLOCK_HZ / 2.0   # wave64 VALU instructions per second: one per 2 cycles and SIMD (plain f32 ops; the guide's rate)
FP32_VECTOR_PEAK = 157.3e12           # MI355X vector fp32 peak, flop/s
FLOP_PER_EVAL_VIEW = 1600.0           # useful flops of one ScorePixelImage evaluation (49 taps x ~32 flop)


def host_cores():
    """the host cores this process can really run on: the scheduler affinity, cut down to the cgroup's CPU quota when there is
    one (a GPU box hands a one-GPU job a share of its cores; spinning on more threads than that only slows the oracle down)"""
    if os.environ.get("HCMVS_CPU_THREADS"):
        return max(1, int(os.environ["HCMVS_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(-(-int(txt[0]) // int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, -(-q // per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(n_threads):
    """The CPU oracle (restatement of the reference algorithm, reference arithmetic) on a bounded sample of
    the same workload: ONE unit of it (1920x1080, 8 source views, 7x7, 8 sweeps), row-pipelined over n_threads host
    threads (about 15 s on 16 cores; `cores` in the result is the thread count actually used = all cores of the box's share)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    synth = importlib.import_module("hc-mvs_amd.synth")
    w, h = W, H
    views = synth.make_views(w, h, FOCAL, N_SRC, seed=2)
    pts = synth.sparse_points(views, 2000)
    L = O.lib()
    ref = O.make_view(views[0])
    d0 = np.zeros((h, w), np.float32); n0 = np.zeros((h, w, 3), np.float32)
    dmin = ctypes.c_float(); dmax = ctypes.c_float()
    L.hcor_splat_init(ctypes.byref(ref), O.fptr(pts), len(pts), O.fptr(d0), O.fptr(n0), ctypes.byref(dmin),
                      ctypes.byref(dmax))
    # a job on a shared box may see more cores than it is given time on (the row-pipelined oracle busy-waits between rows, so
    # oversubscription is costly): a 2-second probe on a quarter-size image picks the thread count that is actually fastest
    visible = n_threads
    cands = sorted({min(visible, c) for c in (8, 16, 32, 64, 128, visible)})
    if len(cands) > 1:
        sv = synth.make_views(480, 270, FOCAL / 4, N_SRC, seed=2)
        sd0 = np.zeros((270, 480), np.float32); sn0 = np.zeros((270, 480, 3), np.float32)
        best = None
        for c in cands:
            pp = O.default_params(adapthalfwin=AHW, n_estimation_iters=1, arith_mode=O.ARITH_REFERENCE, order=O.ORDER_ROWS, n_threads=c)
            t0 = time.time()
            O.estimate(sv, pp, 5.0, 15.0, sd0, sn0)
            dtc = time.time() - t0
            if best is None or dtc < best[0]:
                best = (dtc, c)
            if dtc > 4 * best[0]:
                break
        n_threads = best[1]
    p = O.default_params(adapthalfwin=AHW, n_estimation_iters=SWEEPS, arith_mode=O.ARITH_REFERENCE,
                         order=O.ORDER_ROWS, n_threads=n_threads)
    t0 = time.time()
    O.estimate(views, p, dmin.value, dmax.value, d0, n0)
    dt = time.time() - t0
    return {"value": round(w * h / dt / 1e6, 4), "unit": "Mpix/s", "cores": n_threads, "kind": "port",
            "sample": "one unit of the workload: %dx%d synthetic scene, 8 source views, 7x7 taps, 8 sweeps, one full estimate (%.1f s) on %d threads "
                      "(%d cores visible to the process; the count is the fastest of a short probe)" % (w, h, dt, n_threads, visible)}


def fuse_throughput(ctx, views, pts, dev):
    """Secondary figure of BASELINE.json's metric: FuseDepthMaps (SceneDensify.cpp:3265-3495) points/s on ESTIMATED maps: the
    nine 1080p views of one synthetic scene are each estimated as the reference image against the other eight (7x7, 8 sweeps,
    one batch), then fused (every view has the other eight as neighbours).  Maps and images are resident on the device, the
    cloud is copied back to the host inside the timed call.  raster_order = the reference's pixel order (cloud identical to the
    sequential algorithm's), hashed_order = hcmvs_set_fuse_order(1)."""
    import numpy as np
    import torch
    binding = importlib.import_module("hc-mvs_amd.binding")
    n = len(views)
    HW = H * W
    for i, v in enumerate(views):
        g8 = np.clip(np.rint(v["gray"] * 255), 0, 255).astype(np.uint8)
        ctx.upload_view(9000 + i, v["gray"], v["K"], v["R"], v["C"], bgr=np.stack([g8, g8, g8], -1).copy())
    work = torch.zeros(n, 5 * HW, dtype=torch.float32, device=dev)
    items, rng = [], []
    for i in range(n):
        d0, n0, dmin, dmax = ctx.splat_init(9000 + i, pts)
        work[i, :HW] = torch.from_numpy(d0).reshape(-1).to(dev); work[i, HW:4 * HW] = torch.from_numpy(n0).reshape(-1).to(dev)
        base = work[i].data_ptr()
        items.append(dict(ref_id=9000 + i, src_ids=[9000 + j for j in range(n) if j != i], d_min=dmin, d_max=dmax, d_depth=base,
                          d_normal=base + 4 * HW, d_conf=base + 16 * HW, seed_offset=100 + i))
        rng.append((dmin, dmax))
    p = binding.default_params(adapthalfwin=AHW, n_estimation_iters=SWEEPS, seed=777)
    torch.cuda.synchronize()
    ctx.estimate_batch_device(items, p)
    ctx.synchronize()
    est = work.clone()
    out = {"views": "%d x %dx%d, estimated maps (7x7, 8 sweeps), 8 neighbours each" % (n, W, H)}
    for mode, name in ((0, "raster_order"), (1, "hashed_order")):
        ctx.set_fuse_order(mode)
        best = None
        for _ in range(2):
            work.copy_(est)                                      # fusion mutates the depth maps
            for i in range(n):
                base = work[i].data_ptr()
                ctx.set_depthmap_device(9000 + i, base, base + 4 * HW, base + 16 * HW, rng[i][0], rng[i][1])
                ctx.set_neighbors(9000 + i, [9000 + j for j in range(n) if j != i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = ctx.fuse([9000 + i for i in range(n)], HW * n // 2)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, got["n_points"], got["n_depths"], int(got["n_views"].astype(np.int64).sum()) - int(got["n_points"]))
        dt, npts, ndep, nmerged = best
        out[name] = {"points_per_s": round(npts / dt), "depths_per_s": round(ndep / dt), "ms": round(dt * 1e3, 2), "points": int(npts)}
        if mode == 0:
            # algorithmic bytes per reference pixel = 20 B own + per neighbour 4 B depth + 4 B claim index, + on
            # agreement 12 B normal + 4 B conf + 3 B colour.  The fusion is a chain of >= 12 small launches per image plus the copy of
            # the cloud to the caller's (pageable) host buffers: latency-bound, nowhere near the HBM roofline -- the fraction says so.
            alg = ndep * (20.0 + (n - 1) * 8.0) + nmerged * 19.0
            cloud_bytes = npts * 31                                # xyz 12 + normal 12 + colour 3 + view count 4
            probe = torch.empty(max(cloud_bytes, 1), dtype=torch.uint8, device=dev)
            host = torch.empty(max(cloud_bytes, 1), dtype=torch.uint8)   # pageable, like the caller's arrays
            host.copy_(probe); torch.cuda.synchronize()                  # (first touch of the host pages outside the timing)
            t0 = time.perf_counter(); host.copy_(probe); torch.cuda.synchronize(); d2h = time.perf_counter() - t0
            out["roofline"] = {"kernel": "FuseDepthMaps (fuse_begin / settle / apply / points / compaction kernels, %d images)" % n, "bound": "latency",
                               "achieved": round(alg / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes": int(alg), "ms": round(dt * 1e3, 2),
                               "d2h_cloud_bytes": int(cloud_bytes), "d2h_ms_probe": round(d2h * 1e3, 2), "d2h_share": round(min(1.0, d2h / dt), 3),
                               "note": "SURVEY.md 8d bytes (20 + N_nb x 8 per valid depth, + 19 per merged estimate) over the wall time of hcmvs_fuse incl. the copy of "
                                       "the cloud to pageable host memory (d2h_ms_probe: a copy of the same size timed alone); >= 12 dependent launches per image"}
    ctx.set_fuse_order(0)
    return out


AUTHORS = dict(batch=16, n_src=10, ahw=7, sweeps=3, outer=4, prop_halfwin=5, prop_step=4, photometric_flow=0.26)


def authors_config(ctx, dev, pmc=True):
    """The reference authors' own settings (data/frame_main/resize3/run.py:35-78: --number-views 10, --n-adapthalfwin 7, 4 outer x 3 inner
    sweeps, cross pattern 5 / 4, photometric_flow 0.26) as a batch of 16 reference images of 1920x1080: outer iteration 1 (the first with
    the cross pattern) is timed, after an untimed outer iteration 0 produced its input maps.  This is the TWO + PACK instance of the sweep
    kernel (two sets of eight view groups, idle groups take pairs of their own), which the headline configuration never runs."""
    import numpy as np
    import torch
    binding = importlib.import_module("hc-mvs_amd.binding")
    synth = importlib.import_module("hc-mvs_amd.synth")
    A = AUTHORS
    B, V, HW = A["batch"], A["n_src"], H * W
    work = torch.empty(B, 5 * HW, dtype=torch.float32, device=dev)
    items, keep, scenes = [], [], {}
    for b_ in range(B):
        if b_ % 2 not in scenes:                       # two distinct scenes; every unit has its own copy of images and maps in HBM
            vs = synth.make_views(W, H, FOCAL, V, seed=40 + b_ % 2)
            scenes[b_ % 2] = (vs, synth.sparse_points(vs, 2000, seed=45 + b_ % 2))
        views, pts = scenes[b_ % 2]
        slab = torch.from_numpy(np.stack([v["gray"] for v in views])).to(dev)
        keep.append(slab)
        for i, v in enumerate(views):
            ctx.set_view_device(20000 + 100 * b_ + i, W, H, slab[i].data_ptr(), v["K"], v["R"], v["C"])
        ctx.shapes[20000 + 100 * b_] = (H, W)
        d0, n0, dmin, dmax = ctx.splat_init(20000 + 100 * b_, pts)
        work[b_, :HW] = torch.from_numpy(d0).reshape(-1).to(dev); work[b_, HW:4 * HW] = torch.from_numpy(n0).reshape(-1).to(dev); work[b_, 4 * HW:] = 0
        base = work[b_].data_ptr()
        items.append(dict(ref_id=20000 + 100 * b_, src_ids=[20000 + 100 * b_ + i for i in range(1, V + 1)], d_min=dmin, d_max=dmax, d_depth=base,
                          d_normal=base + 4 * HW, d_conf=base + 16 * HW, seed_offset=b_))
    kw = dict(adapthalfwin=A["ahw"], n_estimation_iters=A["sweeps"], n_external_iters=A["outer"], propagate_halfwin=A["prop_halfwin"],
              propagate_step=A["prop_step"], photometric_flow=A["photometric_flow"], seed=4321)
    torch.cuda.synchronize()
    ctx.estimate_batch_device(items, binding.default_params(it_external=0, **kw))
    ctx.synchronize()
    after0 = work.clone()
    best = None
    for _ in range(2):
        work.copy_(after0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.estimate_batch_device(items, binding.default_params(it_external=1, **kw))
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        if best is None or st.ms_sweep_avg < best[0]:
            best = (st.ms_sweep_avg, dt, st.ms_total, st.evals, st.evals_issued)
    ms_sweep, dt, ms_total, evals, issued = best
    launches = max(1, ctx.stats().n_sweep_launches)
    spl = A["sweeps"] // launches
    P = (W - 14) * (H - 14)
    out = {"workload": "batch of %d reference images x %d source views, 1920x1080, 8x8 taps (adapthalfwin 7), outer iteration 1 of 4 (cross pattern %d / %d), "
                       "%d sweeps, photometric_flow %.2f" % (B, V, A["prop_halfwin"], A["prop_step"], A["sweeps"], A["photometric_flow"]),
           "kernel": "sweep_kernel<8,1,false,TWO,PACK>", "sweep_ms": round(ms_sweep, 3), "sweeps_per_launch": spl, "sweep_launch_ms": round(ms_sweep * spl, 3),
           "Mpix_s_per_sweep": round(B * W * H / ms_sweep / 1e3, 2),
           "estimate_ms": round(ms_total, 2), "Mpix_s_outer_iteration": round(B * W * H / dt / 1e6, 3),
           "evals_per_pixel_sweep": round((evals / (B * P) - 1) / A["sweeps"], 3), "evals_issued_per_pixel_sweep": round((issued / (B * P) - 1) / A["sweeps"], 3)}
    if pmc:
        live = live_pmc(B, kernel="sweep_kernel<8, 1, false, true, true", extra=["--authors-only"], sets=(("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"),),
                        need=("SQ_INSTS_VALU",))
        if live:
            t = ms_sweep * spl * 1e-3
            out["valu"] = {"insts_per_launch": int(live["SQ_INSTS_VALU"]), "insts_per_pixel_sweep": round(live["SQ_INSTS_VALU"] / (B * P * spl), 1),
                           "frac": round(live["SQ_INSTS_VALU"] / t / VALU_PEAK_INSTS, 4),
                           "busy_frac_rocprof_4_cycles": round(live.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (N_SIMD * CLOCK_HZ * t), 4),
                           "counter_source": "rocprofv3 --pmc child run of `bench.py --authors-only`, now"}
        else:
            out["valu"] = None
    for b_ in range(B):
        for i in range(V + 1):
            ctx.release_view(20000 + 100 * b_ + i)
    return out


def copy_bandwidth(dev):
    """measured device copy bandwidth (read + write) of a 1 GiB float4 stream, beside the 8 TB/s spec"""
    import torch
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0); b = torch.empty_like(a)
    best = 0.0
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = max(best, 2.0 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return round(best, 1)


PMC_FILE = "r04_pmc.json"


def pmc_value(batch, what, spl=1):
    """Per-launch counter totals of the sweep kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/r02_pmc.json, made by tools/prof_bench.sh + profiles/summarize_pmc.py): `hbm_bytes` = FETCH_SIZE + WRITE_SIZE,
    `valu_insts` = SQ_INSTS_VALU, `valu_busy_quadcycles` = SQ_ACTIVE_INST_VALU.  PMC counters cannot be collected from inside the
    process (rocprofv3 wraps it), so these are the committed measurement of the same command, not this run's; null for any
    other batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            j = json.load(f)
        v = j.get("sweep_kernel_batch%d_%s_per_launch" % (batch, what))
        return None if v is None else int(v * spl / max(1, int(j.get("sweeps_per_launch", 1))))   # a launch of the file may hold another number of sweeps
    except OSError:
        return None


def live_pmc(batch, timeout_s=150, kernel=None, extra=(), sets=(("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU")),
             need=("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")):
    """HBM traffic and VALU instruction counters of the sweep kernel, measured NOW: one `rocprofv3 --pmc` child run of this same
    command (one step, same batch) per counter set, collected the way MI355X_MICROARCH.md prescribes (separate --pmc passes, no
    trace domain beside them).  The children are ordinary subprocesses; this process only waits.  Returns {counter: mean per
    sweep launch} or None when rocprofv3 cannot be used here (absent, failing, or this process is itself being profiled)."""
    import csv, glob, shutil, subprocess, tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) or k == "HSA_TOOLS_LIB" for k in os.environ):
        return None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    kernel = kernel or "sweep_kernel<8, %d" % (3 if batch == 1 else (2 if batch <= 3 else 1))   # waves per row the library picks for 1080p images
    tmp = tempfile.mkdtemp(prefix="hcmvs_pmc_", dir="/tmp")
    res = {}
    try:
        for i, cs in enumerate(sets):
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc", *cs, "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"), "--steps", "1",
                   "--warmup", "0", "--batch", str(batch), "--no-cpu-baseline", "--no-fuse", "--no-pmc", "--no-authors", *extra]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            acc = {}
            for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel in row["Kernel_Name"]:
                            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for c, v in acc.items():
                res[c] = sum(v) / len(v)
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res if all(k in res for k in need) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="independent reference images per step and GPU (1..64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fuse", action="store_true", help="skip the FuseDepthMaps points/s figure")
    ap.add_argument("--no-authors", action="store_true", help="skip the authors_config figure (10 views, 8x8 taps, cross pattern: the TWO + PACK kernel)")
    ap.add_argument("--authors-only", action="store_true", help="run only the authors_config workload (what its rocprofv3 child run executes)")
    ap.add_argument("--no-pmc", action="store_true", help="take roofline.traffic / the VALU counters from profiles/ instead of measuring them "
                                                          "in rocprofv3 --pmc child runs")
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
    # rehearsal switches (not for measurements): all ranks on device 0 with the gloo backend, to exercise the N > 1 code
    # path on a one-GPU box
    if os.environ.get("HCMVS_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("HCMVS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch with torch.distributed.run for N > 1)"

    binding = importlib.import_module("hc-mvs_amd.binding")  # imports torch first: one HIP runtime per process
    synth = importlib.import_module("hc-mvs_amd.synth")
    D = importlib.import_module("hc-mvs_amd.distributed")

    # synthetic scenes: every rank gets its own batch of B units (weak scaling); unit b = views 100*b .. 100*b+8
    B = args.batch
    ctx = binding.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    if args.authors_only:
        print(json.dumps({"authors_config": authors_config(ctx, dev, pmc=False)}), flush=True)
        ctx.close()
        return
    HW = H * W
    params = binding.default_params(adapthalfwin=AHW, n_estimation_iters=SWEEPS, it_external=0, n_external_iters=1,
                                    seed=1234)
    items, inits, works, keep, scenes = [], [], [], [], {}
    allwork = torch.empty(B, 5 * HW, dtype=torch.float32, device=dev)  # the rank's packed maps: unit b -> allwork[b]
    gathered = torch.empty(world * B, 5 * HW, dtype=torch.float32, device=dev) if world > 1 else None
    for b_ in range(B):
        if b_ % 4 not in scenes:  # 4 distinct scenes per rank; every unit still gets its own copy in HBM
            vs = synth.make_views(W, H, FOCAL, N_SRC, seed=2 + rank * 4 + b_ % 4)
            scenes[b_ % 4] = (vs, synth.sparse_points(vs, 2000, seed=5 + rank * 4 + b_ % 4))
        views, pts = scenes[b_ % 4]
        slab = torch.from_numpy(np.stack([v["gray"] for v in views])).to(dev)  # one allocation per unit
        keep.append(slab)
        for i, v in enumerate(views):
            ctx.set_view_device(100 * b_ + i, W, H, slab[i].data_ptr(), v["K"], v["R"], v["C"])
        ctx.shapes[100 * b_] = (H, W)
        # initial maps (SceneDensify.cpp:783-808 splat of the sparse points), resident on the device
        d0, n0, dmin, dmax = ctx.splat_init(100 * b_, pts)
        init = torch.cat([torch.from_numpy(d0).reshape(-1), torch.from_numpy(n0).reshape(-1), torch.zeros(HW)]).to(dev)
        work = allwork[b_]
        esz = work.element_size()
        inits.append(init); works.append(work)
        items.append(dict(ref_id=100 * b_, src_ids=[100 * b_ + i for i in range(1, N_SRC + 1)], d_min=dmin, d_max=dmax,
                          d_depth=work.data_ptr(), d_normal=work.data_ptr() + HW * esz, d_conf=work.data_ptr() + 4 * HW * esz,
                          seed_offset=b_))

    def step(its=items):
        for init, work in zip(inits, works):
            work.copy_(init)
        ctx.estimate_batch_device(its, params)
        if world > 1:  # the exchange FuseDepthMaps needs: every rank receives every map (20 B/px)
            D.allgather_maps(allwork, out=gathered)

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
    # single-unit latency, outside the timed region and without the exchange
    def single():
        works[0].copy_(inits[0])
        ctx.estimate_batch_device(items[:1], params)
        torch.cuda.synchronize()
    single()
    t1 = time.perf_counter(); single()
    single_ms = (time.perf_counter() - t1) * 1e3
    st1 = ctx.stats()  # the single unit's own kernel times and evaluation counts

    if rank == 0:
        P = (W - 14) * (H - 14)
        taps_a = 0  # pass A scores every pixel of every unit once
        for b_ in range(B):
            gra = ctx.gradient_map(100 * (b_ % 4))[7:H - 7, 7:W - 7]  # same content every 4 units
            taps_a += int(np.where(gra > 100, 36, (AHW + 1) ** 2).astype(np.int64).sum())
        tap_evals_sweeps = int(st.tap_evals) - taps_a
        # algorithmic bytes of ONE sweep launch (SURVEY.md 8d tap-gather convention): every bilinear sample
        # counts its 4 texels (16 B) per source view, plus the 24 B of per-pixel state read and 20 B written
        # a batch of 16 or more images runs all its sweeps in ONE kernel launch (hcmvs_api.cpp): "per launch" is then per 8 sweeps
        spl = SWEEPS // max(1, st.n_sweep_launches)                # sweeps per sweep-kernel launch
        ms_launch = st.ms_sweeps / max(1, st.n_sweep_launches)      # average duration of a sweep-kernel launch (HIP events on its stream)
        bytes_sweep = (tap_evals_sweeps / SWEEPS * N_SRC * 16.0 + B * P * (24.0 + 20.0)) * spl
        achieved = bytes_sweep / (ms_launch * 1e-3) / 1e9
        out = {
            "metric": "PatchMatch Mpix/s (1080p, 8 views, 7x7, 8 iter)",
            "value": round(world * B * W * H * args.steps / dt / 1e6, 4),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "batch of %d independent units per GPU, each 1 ref x 8 src views, 1920x1080, 7x7 taps "
                                   "(adapthalfwin 6), 8 sweeps, it_external 0, full EstimateDepthMap (median + init score + "
                                   "sweeps + end pass)" % B,
                       "units_per_step": "%d reference images per GPU (%d px each)" % (B, W * H), "batch": B,
                       "exchange": ("%s all-gather of 20 B/px maps per step" % ("RCCL" if backend == "nccl" else backend + " (rehearsal, not a measurement)")) if world > 1 else "none",
                       "evals_per_pixel_sweep": round((st.evals / (B * P) - 1) / SWEEPS, 3)},
            "per_gpu": round(B * W * H * args.steps / dt / 1e6, 4),
            "single_unit": {"ms": round(single_ms, 2), "Mpix/s": round(W * H / single_ms / 1e3, 3)},
            "kernel_ms": {"score_pass": round(st.ms_score, 3), "sweep_avg": round(st.ms_sweep_avg, 3), "sweep_launches": int(st.n_sweep_launches),
                          "sweep_launch_avg": round(ms_launch, 3), "sweeps_total": round(st.ms_sweeps, 3), "end": round(st.ms_end, 3),
                          "estimate_total": round(st.ms_total, 3)},
            # SURVEY.md 8d convention, kept for continuity: tap-gather ALGORITHMIC bytes (every bilinear sample counts its 4 texels)
            # over the HBM peak.  The gather is served by L2, so the fraction can pass 1: it says how many taps per second are
            # sampled, not what limits the kernel -- that is `roofline` below (the vector ALU).
            "roofline_convention": {"kernel": "sweep_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                                    "compulsory_bytes_per_launch": int(B * P * (4 * N_SRC + 4 + 40)) * spl,
                                    "algorithmic_bytes_per_launch": int(bytes_sweep), "sweeps_per_launch": spl,
                                    "avg_launch_ms": round(ms_launch, 3),
                                    "convention": "SURVEY.md 8d: algorithmic tap-gather bytes (4 texels per bilinear sample) over the HBM peak"},
        }
        # the counters of the sweep kernel: measured now in rocprofv3 --pmc child runs of this command (FETCH_SIZE / WRITE_SIZE are
        # reported in KiB); the committed measurement of the same command only when that is not possible
        live = live_pmc(B) if world == 1 and not args.no_pmc else None
        if live:
            traffic = int((live["FETCH_SIZE"] + live["WRITE_SIZE"]) * 1024)
            vi, vb = int(live["SQ_INSTS_VALU"]), int(live.get("SQ_ACTIVE_INST_VALU", 0)) or None
            source = "rocprofv3 --pmc child runs of this command, now (SQ_INSTS_VALU + SQ_ACTIVE_INST_VALU, FETCH_SIZE, WRITE_SIZE: one pass per set)"
        else:
            traffic, vi, vb = pmc_value(B, "hbm_bytes", spl), pmc_value(B, "valu_insts", spl), pmc_value(B, "valu_busy_quadcycles", spl)
            source = "profiles/%s (committed rocprofv3 --pmc measurement of this command)" % PMC_FILE
        t_launch = ms_launch * 1e-3
        evals_launch = (st.evals - B * P) / SWEEPS * spl             # ScorePixel evaluations of one sweep-kernel launch (each covers all views)
        useful_flops = evals_launch * N_SRC * FLOP_PER_EVAL_VIEW
        roof = {"kernel": "sweep_kernel<8,1> (batched row worker; %d sweeps per launch)" % spl, "bound": "valu",
                "achieved": round(vi / t_launch, 1) if vi else None, "peak": VALU_PEAK_INSTS, "unit": "wave64 VALU instructions/s",
                "frac": round(vi / t_launch / VALU_PEAK_INSTS, 4) if vi else None,
                "useful_flop_frac": round(useful_flops / t_launch / FP32_VECTOR_PEAK, 4),
                "useful_TFLOPs": round(useful_flops / t_launch / 1e12, 2),
                "traffic": traffic, "avg_launch_ms": round(ms_launch, 3), "sweeps_per_launch": spl, "counter_source": source,
                "peak_definition": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (plain f32 rate, MI355X_MICROARCH.md); useful flops = "
                                   "evaluations x views x 1.6 kflop (SURVEY.md 8d) over the 157.3 TFLOP/s vector fp32 peak"}
        if traffic:
            roof["hbm_frac"] = round(traffic / t_launch / 1e9 / HBM_PEAK_GBS, 5)
        if vi:
            simd_cycles = N_SIMD * CLOCK_HZ * t_launch
            roof["insts_per_launch"] = vi
            roof["insts_per_pixel_sweep"] = round(vi / (B * P * spl), 1)
            # the same instruction count priced at what this mix costs: measured on this chip (profiles/r02_valu_issue_bench.jsonl)
            # fma/mul 2.3-2.8 cycles, logic 3.0, DPP / cvt / mul24 / readlane / f64 4.2-4.7, rcp 8.2 -> about 3 cycles per
            # instruction; and rocprof's own busy counter (SQ_ACTIVE_INST_VALU, 4-cycle units)
            roof["pipe_frac_at_3_cycles"] = round(vi * 3 / simd_cycles, 4)
            roof["busy_frac_rocprof_4_cycles"] = round(vb * 4 / simd_cycles, 4) if vb else None
        out["roofline"] = roof
        out["roofline_convention"]["copy_bandwidth_measured_GBs"] = copy_bandwidth(dev)
        # the same roofline figure for ONE image alone (SURVEY.md 8d's literal `t` = one complete estimate): latency-bound by
        # the (W + H) x T_pixel critical path of its row wavefront
        taps_a1 = int(np.where(ctx.gradient_map(0)[7:H - 7, 7:W - 7] > 100, 36, (AHW + 1) ** 2).astype(np.int64).sum())
        bytes_1 = (int(st1.tap_evals) - taps_a1) / SWEEPS * N_SRC * 16.0 + P * 44.0
        out["roofline_single_unit"] = {"kernel": "sweep_kernel<8,3> (one image alone: 3 waves per row, the rows handed out in stretches of 256 columns; hcmvs_api.cpp waves-per-row policy)",
                                       "bound": "latency of the row wavefront; the figure is in the SURVEY.md 8d convention (hbm)",
                                       "achieved": round(bytes_1 / (st1.ms_sweep_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(bytes_1 / (st1.ms_sweep_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "avg_launch_ms": round(st1.ms_sweep_avg, 3), "estimate_ms": round(st1.ms_total, 2)}
        if world == 1 and not args.no_fuse:
            out["fuse"] = fuse_throughput(ctx, scenes[0][0], scenes[0][1], dev)
        if world == 1 and not args.no_authors:
            out["authors_config"] = authors_config(ctx, dev, pmc=not args.no_pmc)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host_cores())   # every host core this process may use
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
