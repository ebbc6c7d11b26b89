"""GPU parity of the estimation path (pass A, sweeps, end pass) against the CPU oracle, through the C-ABI.

The oracle runs in its device-association arithmetic (same IEEE operation sequence as the kernels), so the
comparison is BIT-EXACT for every map, including all data-dependent control flow (propagation decisions,
random refinement, the two-best-views selection) and the cross-wave hand-off protocol of the sweeps.
"""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

binding = importlib.import_module("hc-mvs_amd.binding")
synth = importlib.import_module("hc-mvs_amd.synth")


@pytest.fixture(scope="module")
def ctx():
    c = binding.Context(0)
    yield c
    c.close()


def _scene(w, h, f, n_src, seed, n_pts=80):
    views = synth.make_views(w, h, f, n_src, seed=seed)
    pts = synth.sparse_points(views, n_pts)
    return views, pts


def _upload(ctx, views):
    for i, v in enumerate(views):
        ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])


def _params(**kw):
    pg = binding.default_params(**kw)
    po = O.default_params(arith_mode=O.ARITH_DEVICE, order=O.ORDER_ROWS, n_threads=8, **kw)
    return pg, po


def _compare(got, want):
    names = ("depth", "normal", "conf")
    for g, w, n in zip(got, want, names):
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)
            raise AssertionError("%s differs at %d elements, first %s: gpu %r oracle %r" %
                                 (n, len(bad), bad[0], g[tuple(bad[0])], w[tuple(bad[0])]))


def test_gradient_map_matches_oracle(ctx):
    views, _ = _scene(96, 72, 90.0, 1, seed=4)
    _upload(ctx, views)
    assert np.array_equal(ctx.gradient_map(0), O.gradient_map(views[0]["gray"]))


@pytest.mark.parametrize("n_src,ahw", [(1, 6), (2, 5), (3, 6), (4, 7), (5, 6), (8, 6), (8, 7), (9, 5), (12, 6)])
def test_estimate_bit_exact(ctx, n_src, ahw):
    views, pts = _scene(112, 88, 100.0, n_src, seed=10 + n_src)
    _upload(ctx, views)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    pg, po = _params(adapthalfwin=ahw, n_estimation_iters=3, seed=77 + n_src)
    got = ctx.estimate(0, list(range(1, n_src + 1)), pg, dmin, dmax, d0, n0)
    st = ctx.stats()
    do, no, co, ev = O.estimate(views, po, dmin, dmax, d0, n0)
    assert st.evals == ev
    _compare(got, (do, no, co))
    assert (got[0] > 0).mean() > 0.3


@pytest.mark.parametrize("n_src,ahw", [(8, 10), (3, 9), (1, 8), (2, 10), (10, 10), (6, 8)])
def test_estimate_bit_exact_beyond_64_taps(ctx, n_src, ahw):
    """patches beyond the reference's nTexels = 64 (DepthMap.h:354-358 generalised; BASELINE.json configs[4] uses 11 x 11):
    half windows 8..10, every lane layout, border = half window; pixels with a strong gradient still use the 6 x 6 patch"""
    views, pts = _scene(120, 96, 100.0, n_src, seed=80 + n_src + ahw)
    _upload(ctx, views)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    pg, po = _params(adapthalfwin=ahw, n_estimation_iters=2, seed=7 + ahw)
    got = ctx.estimate(0, list(range(1, n_src + 1)), pg, dmin, dmax, d0, n0)
    st = ctx.stats()
    do, no, co, ev = O.estimate(views, po, dmin, dmax, d0, n0)
    assert st.evals == ev
    _compare(got, (do, no, co))
    assert (got[0][:ahw] == 0).all() and (got[0][:, :ahw] == 0).all() and (got[0][ahw:-ahw, ahw:-ahw] > 0).mean() > 0.3
    gra = ctx.gradient_map(0)[ahw:-ahw, ahw:-ahw]
    assert 0.02 < (gra > 100).mean() < 0.98     # both patch sizes occur


def test_big_patch_cross_pattern_in_a_batch(ctx):
    """11 x 11 patch + outer-iteration cross pattern + batch of two images with two waves per row"""
    torch = pytest.importorskip("torch")
    scenes = [_scene(104, 88, 95.0, 5, seed=91), _scene(88, 104, 95.0, 5, seed=92)]
    pg, po = _params(adapthalfwin=10, n_estimation_iters=2, seed=17, it_external=1, n_external_iters=3, propagate_halfwin=5, propagate_step=2)
    got, keep = _batch_run(ctx, torch, scenes, pg, [0, 3])
    for si, ((views, pts), g, k) in enumerate(zip(scenes, got, keep)):
        po.seed = 17 + [0, 3][si]
        _compare(g, O.estimate(views, po, k[5], k[6], k[3], k[4])[:3])


def test_restore_variant_hint_hypothesis(ctx):
    """the fork's `restore` variant (restore/libs/MVS/DepthMap.cpp:1527-1549): in the last sweep of the last outer iteration
    every pixel also tries the estimate of the up-sampled coarser level and takes it when it is at most 0.1 worse"""
    torch = pytest.importorskip("torch")
    import ctypes as C
    dev = torch.device("cuda:0")
    views, pts = _scene(120, 96, 100.0, 3, seed=47)
    _upload(ctx, views)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    rng = np.random.RandomState(3)
    hint_d = (views[0]["depth"] * (1 + 0.004 * rng.normal(size=views[0]["depth"].shape))).astype(np.float32)
    hint_d[rng.uniform(size=hint_d.shape) < 0.2] = 0          # holes in the coarser level: no extra hypothesis there
    hint_n = np.ascontiguousarray(views[0]["normal"], np.float32)
    res = {}
    # "widened": as the restore variant really runs -- its depth range takes the enlarged coarser map in, zeros included
    # (restore/libs/MVS/SceneDensify.cpp:526-532), so the lower bound is 0
    for use_hint in (False, True, "widened"):
        lo = 0.0 if use_hint == "widened" else dmin
        for it in (0, 1):                                     # two outer iterations: the hint acts in the last sweep of the last one
            pg, po = _params(adapthalfwin=6, n_estimation_iters=2, seed=5, it_external=it, n_external_iters=2, propagate_halfwin=5, propagate_step=4)
            if it == 0:
                g = (d0, n0, np.zeros_like(d0)); o = (d0, n0)
            td = torch.from_numpy(g[0]).to(dev); tn = torch.from_numpy(g[1]).to(dev); tc = torch.from_numpy(g[2]).to(dev)
            hd = torch.from_numpy(hint_d).to(dev); hn = torch.from_numpy(hint_n).to(dev)
            item = dict(ref_id=0, src_ids=[1, 2, 3], d_min=lo, d_max=dmax, d_depth=td.data_ptr(), d_normal=tn.data_ptr(), d_conf=tc.data_ptr())
            if use_hint:
                item.update(d_hint_depth=hd.data_ptr(), d_hint_normal=hn.data_ptr())
                po.hint_depth = hint_d.ctypes.data_as(C.POINTER(C.c_float)); po.hint_normal = hint_n.ctypes.data_as(C.POINTER(C.c_float))
            torch.cuda.synchronize()
            ctx.estimate_batch_device([item], pg)
            ctx.synchronize()
            g = (td.cpu().numpy(), tn.cpu().numpy(), tc.cpu().numpy())
            want = O.estimate(views, po, lo, dmax, o[0], o[1])
            _compare(g, want[:3])
            assert ctx.stats().evals == want[3]
            o = (want[0], want[1])
        res[use_hint] = g
    changed = (res[True][0] != res[False][0]).mean()
    assert 0.05 < changed < 0.9                              # the hint replaced a good part of the estimates, not all
    gt = views[0]["depth"]
    acc = {k: float((np.abs(v[0] - gt)[v[0] > 0] / gt[v[0] > 0] < 0.01).mean()) for k, v in res.items()}
    assert acc[True] >= acc[False] - 0.01 and acc["widened"] >= acc[False] - 0.02   # an accurate coarser level does not hurt


def test_outer_iterations_cross_pattern(ctx):
    """it_external >= 1 uses the cross propagation pattern (DepthMap.cpp:1064-1274) and no end pass until
    the last outer iteration; maps are handed from one call to the next."""
    views, pts = _scene(120, 96, 110.0, 3, seed=21)
    _upload(ctx, views)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    g = (d0, n0, None)
    o = (d0, n0, None)
    for it in range(3):
        pg, po = _params(adapthalfwin=6, n_estimation_iters=2, it_external=it, n_external_iters=3,
                         propagate_halfwin=5, propagate_step=2)
        g = ctx.estimate(0, [1, 2, 3], pg, dmin, dmax, g[0], g[1], g[2])
        od, on, oc, _ = O.estimate(views, po, dmin, dmax, o[0], o[1])
        o = (od, on, oc)
        _compare(g, o)
    assert (g[0] > 0).mean() > 0.3


def test_host_and_device_paths_agree(ctx):
    torch = pytest.importorskip("torch")
    views, pts = _scene(96, 80, 90.0, 2, seed=5)
    _upload(ctx, views)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    pg, _ = _params(adapthalfwin=6, n_estimation_iters=2)
    want = ctx.estimate(0, [1, 2], pg, dmin, dmax, d0, n0)
    dev = torch.device("cuda:0")
    td = torch.from_numpy(d0).to(dev); tn = torch.from_numpy(n0).to(dev); tc = torch.zeros_like(td)
    torch.cuda.synchronize()
    ctx.estimate_device(0, [1, 2], pg, dmin, dmax, td.data_ptr(), tn.data_ptr(), tc.data_ptr())
    ctx.synchronize()
    _compare((td.cpu().numpy(), tn.cpu().numpy(), tc.cpu().numpy()), want)


def test_errors_are_reported(ctx):
    views, pts = _scene(96, 80, 90.0, 1, seed=6)
    _upload(ctx, views)
    d0, n0, dmin, dmax = ctx.splat_init(0, pts)
    pg, _ = _params()
    with pytest.raises(binding.HcmvsError) as e:
        ctx.estimate(0, [99], pg, dmin, dmax, d0, n0)
    assert e.value.code == binding.ERR_INVALID and "99" in str(e.value)
    pg.adapthalfwin = 11                      # the generalised patch goes up to 11 x 11 taps (half window 10)
    with pytest.raises(binding.HcmvsError):
        ctx.estimate(0, [1], pg, dmin, dmax, d0, n0)


def test_batch_of_reference_images_bit_exact(ctx):
    """hcmvs_estimate_batch_device: rows of several reference images interleaved in one sweep launch; every item must
    equal its own single estimate (and therefore the oracle) bit for bit, with different image sizes in one batch."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    scenes = [_scene(112, 88, 100.0, 3, seed=31), _scene(96, 120, 100.0, 3, seed=32), _scene(112, 88, 100.0, 3, seed=33)]
    items, keep, wants = [], [], []
    vid = 0
    for si, (views, pts) in enumerate(scenes):
        ids = list(range(vid, vid + len(views)))
        for i, v in zip(ids, views):
            ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
        vid += len(views)
        ref = ids[si % len(ids)]                      # a different view of each scene is the reference
        srcs = [i for i in ids if i != ref]
        d0, n0, dmin, dmax = ctx.splat_init(ref, pts)
        pg, po = _params(adapthalfwin=6, n_estimation_iters=3, seed=500)
        po.seed = 500 + 7 * si
        ordered = [views[ref - ids[0]]] + [views[i - ids[0]] for i in srcs]
        wants.append(O.estimate(ordered, po, dmin, dmax, d0, n0))
        td = torch.from_numpy(d0).to(dev); tn = torch.from_numpy(n0).to(dev); tc = torch.zeros_like(td)
        keep.append((td, tn, tc))
        items.append(dict(ref_id=ref, src_ids=srcs, d_min=dmin, d_max=dmax, d_depth=td.data_ptr(), d_normal=tn.data_ptr(),
                          d_conf=tc.data_ptr(), seed_offset=7 * si))
    torch.cuda.synchronize()
    ctx.estimate_batch_device(items, pg)
    ctx.synchronize()
    st = ctx.stats()
    assert st.evals == sum(w[3] for w in wants)
    for (td, tn, tc), w in zip(keep, wants):
        _compare((td.cpu().numpy(), tn.cpu().numpy(), tc.cpu().numpy()), w[:3])


def _batch_run(ctx, torch, scenes, pg, seed_offsets, refs=None):
    """estimate every scene's view 0 (or refs[i]) in ONE batch call; returns the device maps as numpy triples"""
    dev = torch.device("cuda:0")
    items, keep = [], []
    vid = 1000
    for si, (views, pts) in enumerate(scenes):
        ids = list(range(vid, vid + len(views)))
        for i, v in zip(ids, views):
            ctx.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
        vid += len(views)
        ref = ids[0]
        d0, n0, dmin, dmax = ctx.splat_init(ref, pts)
        td = torch.from_numpy(d0).to(dev); tn = torch.from_numpy(n0).to(dev); tc = torch.zeros_like(td)
        keep.append((td, tn, tc, d0, n0, dmin, dmax))
        items.append(dict(ref_id=ref, src_ids=ids[1:], d_min=dmin, d_max=dmax, d_depth=td.data_ptr(), d_normal=tn.data_ptr(),
                          d_conf=tc.data_ptr(), seed_offset=seed_offsets[si]))
    torch.cuda.synchronize()
    ctx.estimate_batch_device(items, pg)
    ctx.synchronize()
    return [(k[0].cpu().numpy(), k[1].cpu().numpy(), k[2].cpu().numpy()) for k in keep], keep


def test_batch_more_than_eight_views_and_cross_pattern(ctx):
    """9..16 source views (two sets of eight view groups, pair packing in the partly filled one) and the outer-iteration cross
    pattern inside a batch: every item equals the oracle bit for bit"""
    torch = pytest.importorskip("torch")
    scenes = [_scene(88, 72, 90.0, 10, seed=41), _scene(88, 72, 90.0, 10, seed=42), _scene(72, 88, 90.0, 10, seed=43)]
    pg, po = _params(adapthalfwin=6, n_estimation_iters=2, seed=77, it_external=1, n_external_iters=3, propagate_halfwin=5,
                     propagate_step=2)
    got, keep = _batch_run(ctx, torch, scenes, pg, [0, 5, 9])
    for si, ((views, pts), g, k) in enumerate(zip(scenes, got, keep)):
        po.seed = 77 + [0, 5, 9][si]
        want = O.estimate(views, po, k[5], k[6], k[3], k[4])
        _compare(g, want[:3])


@pytest.mark.parametrize("seg,nw", [("32", "1"), ("40", "2"), ("32", "3")])
def test_rows_handed_out_in_stretches(seg, nw):
    """the sweep worker's tickets as stretches of a row (hcmvs_api.cpp: batches whose rows do not all fit the chip; here forced on small
    images with HCMVS_SWEEP_SEGMENT): a stretch waits for the one to its left, re-reads the ring of the row's latest results from memory
    and goes on where that one stopped -- every map equals the oracle bit for bit, with the plain and the cross propagation pattern,
    8 and 10 source views, odd sweeps (right-to-left), one to three waves per row, stretches that do not divide the row"""
    torch = pytest.importorskip("torch")
    import os
    os.environ["HCMVS_SWEEP_SEGMENT"] = seg
    os.environ["HCMVS_WAVES_PER_ROW"] = nw
    os.environ["HCMVS_SWEEP_LAUNCHES"] = "per-sweep"
    try:
        c = binding.Context(0)
    finally:
        for k in ("HCMVS_SWEEP_SEGMENT", "HCMVS_WAVES_PER_ROW", "HCMVS_SWEEP_LAUNCHES"):
            os.environ.pop(k, None)
    try:
        scenes = [_scene(136, 72, 100.0, 8, seed=91), _scene(88, 104, 100.0, 3, seed=92), _scene(120, 80, 100.0, 8, seed=93)]
        pg, po = _params(adapthalfwin=6, n_estimation_iters=3, seed=611)
        got, keep = _batch_run(c, torch, scenes, pg, [0, 3, 4])
        evals = 0
        for si, ((views, pts), g, k) in enumerate(zip(scenes, got, keep)):
            po.seed = 611 + [0, 3, 4][si]
            want = O.estimate(views, po, k[5], k[6], k[3], k[4])
            _compare(g, want[:3])
            evals += want[3]
        assert c.stats().evals == evals
        # 10 source views (two sets of view groups), outer iteration 1: the cross pattern reaches 5 pixels back into the ring
        scenes = [_scene(104, 72, 90.0, 10, seed=94), _scene(72, 104, 90.0, 10, seed=95)]
        pg, po = _params(adapthalfwin=7, n_estimation_iters=3, seed=78, it_external=1, n_external_iters=3, propagate_halfwin=5, propagate_step=4)
        got, keep = _batch_run(c, torch, scenes, pg, [0, 6])
        for si, ((views, pts), g, k) in enumerate(zip(scenes, got, keep)):
            po.seed = 78 + [0, 6][si]
            _compare(g, O.estimate(views, po, k[5], k[6], k[3], k[4])[:3])
        # patches beyond 64 taps (11 x 11: the big-patch worker, border 10)
        scenes = [_scene(120, 88, 100.0, 3, seed=96), _scene(96, 104, 100.0, 4, seed=97)]
        pg, po = _params(adapthalfwin=10, n_estimation_iters=2, seed=35)
        got, keep = _batch_run(c, torch, scenes, pg, [0, 2])
        for si, ((views, pts), g, k) in enumerate(zip(scenes, got, keep)):
            po.seed = 35 + [0, 2][si]
            _compare(g, O.estimate(views, po, k[5], k[6], k[3], k[4])[:3])
    finally:
        c.close()


def test_ragged_and_minimum_sizes(ctx):
    """odd image sizes down to the smallest image that still has pixels inside the 7 px border"""
    with pytest.raises(binding.HcmvsError):   # nothing left inside the border
        ctx.upload_view(0, np.zeros((15, 15), np.float32), np.eye(3), np.eye(3), np.zeros(3))
    for (w, h, seed) in ((16, 16, 51), (17, 31, 52), (129, 16, 53)):
        views, pts = _scene(w, h, 60.0, 2, seed=seed, n_pts=10)
        _upload(ctx, views)
        d0, n0, dmin, dmax = ctx.splat_init(0, pts)
        pg, po = _params(adapthalfwin=6, n_estimation_iters=2, seed=9)
        got = ctx.estimate(0, [1, 2], pg, dmin, dmax, d0, n0)
        want = O.estimate(views, po, dmin, dmax, d0, n0)
        _compare(got, want[:3])
        assert (got[0][:7] == 0).all() and (got[0][:, :7] == 0).all()   # the fixed border stays empty (DepthMap.cpp:442-447)


def test_full_size_schedule_invariance():
    """BASELINE.json configs[1] at full size (1920x1080, 8 source views, 7x7, 8 sweeps): the oracle cannot run this in seconds, so
    parity is shown through properties that do not depend on the size -- the maps must not depend on how the rows are
    scheduled (one image alone with two waves per row and whole rows == the same image inside a batch with one wave per row, XCD
    affinity and the rows handed out in stretches of 256 columns), the evaluation count per pixel-sweep is the algorithm's
    (2 propagations + 6 refinements, fewer where a neighbour is already good), and the result converges to the analytic ground truth."""
    torch = pytest.importorskip("torch")
    import os
    os.environ["HCMVS_SWEEP_SEGMENT"] = "256"     # (the automatic policy takes stretches for 6 .. 11 images of this size)
    try:
        c = binding.Context(0)
    finally:
        os.environ.pop("HCMVS_SWEEP_SEGMENT", None)
    c1 = binding.Context(0)
    try:
        W, H = 1920, 1080
        scenes = [_scene(W, H, 1600.0, 8, seed=61, n_pts=2000), _scene(W, H, 1600.0, 8, seed=62, n_pts=2000),
                  _scene(W, H, 1600.0, 8, seed=63, n_pts=2000)]
        SWEEPS = 8                      # the configuration BASELINE.json's metric is quoted on
        pg = binding.default_params(adapthalfwin=6, n_estimation_iters=SWEEPS, seed=4321)
        got3, keep = _batch_run(c, torch, scenes, pg, [0, 1, 2])
        evals3 = c.stats().evals
        # item 1 alone (two waves per row, whole rows, no interleaving), in a context of its own
        views, pts = scenes[1]
        k = keep[1]
        pg1 = binding.default_params(adapthalfwin=6, n_estimation_iters=SWEEPS, seed=4321 + 1)
        ids = list(range(1000 + 9, 1000 + 18))
        for i, v in zip(ids, views):
            c1.upload_view(i, v["gray"], v["K"], v["R"], v["C"])
        alone = c1.estimate(ids[0], ids[1:], pg1, k[5], k[6], k[3], k[4])
        evals1 = c1.stats().evals
        for g, a, n in zip(got3[1], alone, ("depth", "normal", "conf")):
            assert np.array_equal(g, a), n
        P = (W - 14) * (H - 14)
        per_px_sweep = (evals1 / P - 1) / SWEEPS
        assert 6.5 < per_px_sweep <= 8.0
        assert abs(evals3 / 3 - evals1) / evals1 < 0.05
        d = alone[0]
        valid = d > 0
        gt = views[0]["depth"]
        assert valid.mean() > 0.9
        assert (np.abs(d - gt)[valid] / gt[valid] < 0.01).mean() > 0.95
    finally:
        c.close()
        c1.close()


def test_batch_mixed_view_counts(ctx):
    """items with 5, 8, 7 and 3 source views share one launch (one lane layout class up to 8 views: idle view groups of an item
    take (hypothesis, view) pairs of their own); 9 and 8 views do not"""
    torch = pytest.importorskip("torch")
    scenes = [_scene(96, 80, 90.0, 5, seed=71), _scene(96, 80, 90.0, 8, seed=72), _scene(80, 96, 90.0, 7, seed=73), _scene(88, 72, 90.0, 3, seed=76)]
    pg, po = _params(adapthalfwin=6, n_estimation_iters=2, seed=33)
    got, keep = _batch_run(ctx, torch, scenes, pg, [0, 1, 2, 3])
    for si, ((views, pts), g, k) in enumerate(zip(scenes, got, keep)):
        po.seed = 33 + si
        _compare(g, O.estimate(views, po, k[5], k[6], k[3], k[4])[:3])
    with pytest.raises(binding.HcmvsError):
        _batch_run(ctx, torch, [_scene(96, 80, 90.0, 9, seed=74), _scene(96, 80, 90.0, 8, seed=75)], pg, [0, 1])


def test_matches_committed_golden(ctx):
    """the kernels against the committed golden fixture of the device association (tests/golden/, generated by
    make_golden.py from the oracle): depth, normal, score maps and the evaluation count, bit for bit"""
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gdir, "estimate_96x80_v3.npz"))
    gd = np.load(os.path.join(gdir, "estimate_96x80_v3_device.npz"))
    for i in range(len(g["gray"])):
        ctx.upload_view(i, g["gray"][i], g["K"][i], g["R"][i], g["C"][i])
    pg = binding.default_params(adapthalfwin=6, n_estimation_iters=3, seed=int(g["seed"]))
    got = ctx.estimate(0, list(range(1, len(g["gray"]))), pg, float(g["dmin"]), float(g["dmax"]), g["d0"], g["n0"])
    assert ctx.stats().evals == int(gd["evals"])
    _compare(got, (gd["depth"], gd["normal"], gd["conf"]))
