"""Scene-level oracle harness (test infrastructure only): the reference's outer loop over all images of a scene
(Scene::ComputeDepthMaps, SceneDensify.cpp:3684-3716 + the event loop Scene::DenseReconstructionEstimate, :3831-4006) played with
the CPU oracle's per-image functions (hcor_estimate, hcor_postfilter, hcor_fuse_depthmaps).

Two schedules of the post-filters of outer iterations 1 and 2:

  interleave=True   the reference's single-thread event order (SceneDensify.cpp:3889-3965): EVTEstimateDepthMap(k) queues
                    EVTOptimizeDepthMap(k) FIRST, so image k is filtered right after its estimate -- its RemoveSmallSegments fuses against
                    the images > k as the previous outer iteration left them and zeroes depths in them before they are estimated again
  interleave=False  the batch schedule of the device path: every image of the outer iteration is estimated,
                    then the images are filtered one after the other in the same (ascending id) order

Image ids must be 0 .. n-1.  Seeds follow hc-mvs_amd/distributed.py::densify_scene (params.seed + image id)."""
import ctypes as C

import numpy as np

import oracle_lib as O


def gradient_map(view):
    """SceneDensify.cpp:581-595 InitGraMap as the C-ABI computes it: from the colour image when there is one, else from round(gray * 255)"""
    if view.get("bgr") is None:
        return O.gradient_map(view["gray"])
    L = O.lib()
    b = np.ascontiguousarray(view["bgr"], np.uint8)
    h, w = b.shape[:2]
    g8 = np.empty((h, w), np.uint8); gra = np.empty((h, w), np.uint8)
    L.hcor_bgr2gray_u8(O.u8ptr(b), w, h, O.u8ptr(g8))
    L.hcor_gradient_map(O.u8ptr(g8), w, h, O.u8ptr(gra))
    return gra


def densify(views, srcs, neighbors, order, init, n_external_iters=1, postfilter=False, interleave=False, mode=O.ARITH_DEVICE, seed=1234,
            n_threads=8, fuse=True, pf_kw=None, fuse_kw=None, hints=None, **est_kw):
    """views: {id: dict(gray, K, R, C[, bgr])}; srcs / neighbors: {id: [ids]}; order: fusion order; init: {id: (depth0, normal0, d_min, d_max)};
    hints: optional {id: (hint_depth, hint_normal)} (the `restore` variant's extra hypothesis, last sweep of the last outer iteration);
    est_kw: oracle estimate parameters (adapthalfwin, n_estimation_iters, propagate_halfwin, ...).
    Returns dict(maps={id: (depth, normal, conf)}, cloud=..., filled=[...], evals=int)."""
    ids = sorted(views)
    assert ids == list(range(len(ids)))
    gra = {i: gradient_map(views[i]) for i in ids}
    cur = {}
    for i in ids:
        d0, n0, dmin, dmax = init[i]
        cur[i] = dict(K=views[i]["K"], R=views[i]["R"], C=views[i]["C"], depth=np.ascontiguousarray(d0, np.float32).copy(),
                      normal=np.ascontiguousarray(n0, np.float32).copy(), conf=np.zeros(d0.shape, np.float32), bgr=views[i].get("bgr"),
                      d_min=float(dmin), d_max=float(dmax), neighbors=[n for n in neighbors[i] if n in views][:31])
    filled, evals = [], 0

    def estimate(i, it):
        nonlocal evals
        kw = dict(est_kw)
        keep = []
        if hints is not None and i in hints and it == n_external_iters - 1:
            hd = np.ascontiguousarray(hints[i][0], np.float32); hn = np.ascontiguousarray(hints[i][1], np.float32)
            keep += [hd, hn]
            kw["hint_depth"] = O.fptr(hd); kw["hint_normal"] = O.fptr(hn)
        p = O.default_params(arith_mode=mode, order=O.ORDER_ROWS, n_threads=n_threads, it_external=it, n_external_iters=n_external_iters,
                             seed=(seed + i) & 0xFFFFFFFF, **kw)
        vs = [views[i]] + [views[s] for s in srcs[i]]
        d, n, c, ev = O.estimate(vs, p, cur[i]["d_min"], cur[i]["d_max"], cur[i]["depth"], cur[i]["normal"], gra=gra[i])
        cur[i]["depth"], cur[i]["normal"], cur[i]["conf"] = d, n, c
        evals += ev

    def post(i):
        dd, nd, cd, nf = O.postfilter([cur[k] for k in ids], i, gra[i], order, mode=mode, **(pf_kw or {}))
        for k in ids:
            cur[k]["depth"] = dd[k]
        cur[i]["normal"], cur[i]["conf"] = nd, cd
        filled.append(nf)

    for it in range(n_external_iters):
        filt = postfilter and it in (1, 2)
        if filt and interleave:
            for i in ids:
                estimate(i, it)
                post(i)
        else:
            for i in ids:
                estimate(i, it)
            if filt:
                for i in ids:
                    post(i)
    out = dict(maps={i: (cur[i]["depth"].copy(), cur[i]["normal"].copy(), cur[i]["conf"].copy()) for i in ids}, filled=filled, evals=evals)
    if fuse:
        h, w = cur[ids[0]]["depth"].shape
        out["cloud"] = O.fuse_depthmaps([cur[k] for k in ids], list(order), h * w * len(ids) // 2 + 16, **(fuse_kw or {}))
    return out


def splat(view, pts):
    """hcor_splat_init (SceneDensify.cpp:783-808): (depth0, normal0, d_min, d_max)"""
    h, w = view["gray"].shape
    ref = O.make_view(view)
    d0 = np.zeros((h, w), np.float32); n0 = np.zeros((h, w, 3), np.float32)
    lo = C.c_float(); hi = C.c_float()
    O.lib().hcor_splat_init(C.byref(ref), O.fptr(np.ascontiguousarray(pts, np.float32)), len(pts), O.fptr(d0), O.fptr(n0), C.byref(lo), C.byref(hi))
    return d0, n0, lo.value, hi.value


def ring_scene(n=6, w=160, h=128, f=150.0, seed=31, n_src=3, n_points=120):
    """n views of one synthetic scene, every one of them a reference image: (views, srcs, neighbors, order, init) as densify() and
    hc-mvs_amd/distributed.py::densify_scene take them (source views / neighbours = the closest cameras; splat initialisation)"""
    import importlib
    synth = importlib.import_module("hc-mvs_amd.synth")
    base = synth.make_views(w, h, f, n - 1, seed=seed, baseline=(0.04, 0.09))
    views = {i: dict(gray=v["gray"], K=v["K"], R=v["R"], C=v["C"], depth=v["depth"],
                     bgr=np.stack([np.clip(np.rint(v["gray"] * 255), 0, 255).astype(np.uint8)] * 3, -1).copy()) for i, v in enumerate(base)}
    near = {i: [j for j in sorted(range(n), key=lambda j: np.linalg.norm(base[j]["C"] - base[i]["C"])) if j != i] for i in range(n)}
    srcs = {i: near[i][:n_src] for i in range(n)}
    order = list(range(n))
    init = {i: splat(base[i], synth.sparse_points([base[i]], n_points, seed=40 + i)) for i in range(n)}
    return views, srcs, near, order, init


class OracleContext:
    """Stand-in for hc-mvs_amd/binding.py::Context with the oracle as its engine and host memory as "device" memory: lets the CPU tests
    drive the control flow of the multi-rank scene path (sharding, all-gathers, broadcasts, copy-backs) without a GPU.  Device
    association arithmetic, so what it computes is what the real context computes."""

    def __init__(self):
        self.views, self.maps, self.nbrs, self.gra = {}, {}, {}, {}

    @staticmethod
    def _arr(ptr, n):
        return np.ctypeslib.as_array((C.c_float * n).from_address(ptr))

    def upload_view(self, vid, gray, K, R, Cc, bgr=None):
        self.views[vid] = dict(gray=gray, K=K, R=R, C=Cc, bgr=bgr)
        self.gra[vid] = gradient_map(self.views[vid])

    def set_depthmap_device(self, vid, d, n, c, d_min, d_max):
        hw = (self.views[vid]["bgr"] if self.views[vid]["gray"] is None else self.views[vid]["gray"]).shape[:2]
        self.maps[vid] = (d, n, c, d_min, d_max, hw)

    def set_neighbors(self, vid, ids):
        self.nbrs[vid] = list(ids)

    def synchronize(self):
        pass

    def estimate_batch_device(self, items, p):
        for it in items:
            i = it["ref_id"]
            h, w = self.views[i]["gray"].shape
            d = self._arr(it["d_depth"], h * w).reshape(h, w); n = self._arr(it["d_normal"], 3 * h * w).reshape(h, w, 3)
            c = self._arr(it["d_conf"], h * w).reshape(h, w)
            kw = {k: getattr(p, k) for k, _ in p._fields_ if k not in ("seed",)}
            po = O.default_params(arith_mode=O.ARITH_DEVICE, order=O.ORDER_ROWS, n_threads=4, seed=(p.seed + it.get("seed_offset", 0)) & 0xFFFFFFFF, **kw)
            vs = [self.views[i]] + [self.views[s] for s in it["src_ids"]]
            assert all(v["gray"] is not None for v in vs), "estimate with a fuse-only view"
            dd, nn, cc, _ = O.estimate(vs, po, it["d_min"], it["d_max"], d, n, gra=self.gra[i])
            d[...] = dd; n[...] = nn; c[...] = cc

    def _dicts(self):
        ids = sorted(self.maps)
        assert ids == list(range(len(ids)))
        out = []
        for i in ids:
            d, n, c, lo, hi, (h, w) = self.maps[i]
            v = self.views[i]
            out.append(dict(K=v["K"], R=v["R"], C=v["C"], depth=self._arr(d, h * w).reshape(h, w), normal=self._arr(n, 3 * h * w).reshape(h, w, 3),
                            conf=self._arr(c, h * w).reshape(h, w), bgr=v["bgr"], d_min=lo, d_max=hi, neighbors=self.nbrs[i]))
        return out

    def postfilter(self, vid, order, **kw):
        cur = self._dicts()
        dd, nd, cd, nf = O.postfilter(cur, vid, self.gra[vid], list(order), mode=O.ARITH_DEVICE)
        for i, m in enumerate(cur):
            m["depth"][...] = dd[i]
        cur[vid]["normal"][...] = nd; cur[vid]["conf"][...] = cd
        return nf

    def postfilter_sequence(self, vids, order, **kw):
        return sum(self.postfilter(v, order) for v in vids)

    def fuse(self, order, capacity, **kw):
        cur = self._dicts()
        f = O.fuse_depthmaps(cur, list(order), capacity)
        for i, m in enumerate(cur):
            m["depth"][...] = f["depths"][i]
        return f
