"""ctypes binding of the CPU oracle (oracle/libhcmvs_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libhcmvs_oracle.so")

ARITH_REFERENCE, ARITH_DEVICE = 0, 1
ORDER_ZIGZAG, ORDER_ROWS = 0, 1


class View(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("gray", C.POINTER(C.c_float)),
                ("bgr", C.POINTER(C.c_uint8)), ("K", C.c_double * 9), ("R", C.c_double * 9), ("C", C.c_double * 3)]


class Params(C.Structure):
    _fields_ = [("adapthalfwin", C.c_int), ("n_estimation_iters", C.c_int), ("it_external", C.c_int),
                ("n_external_iters", C.c_int), ("propagate_halfwin", C.c_int), ("propagate_step", C.c_int),
                ("n_random_iters", C.c_int), ("ncc_threshold_keep", C.c_float), ("random_depth_ratio", C.c_float),
                ("random_angle1_deg", C.c_float), ("random_angle2_deg", C.c_float),
                ("random_smooth_depth", C.c_float), ("random_smooth_normal_deg", C.c_float),
                ("random_smooth_bonus", C.c_float), ("photometric_flow", C.c_float), ("seed", C.c_uint32),
                ("arith_mode", C.c_int), ("order", C.c_int), ("n_threads", C.c_int), ("median_blur", C.c_int),
                ("hint_depth", C.POINTER(C.c_float)), ("hint_normal", C.POINTER(C.c_float))]


class DepthMap(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("K", C.c_double * 9), ("R", C.c_double * 9),
                ("C", C.c_double * 3), ("depth", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)),
                ("conf", C.POINTER(C.c_float)), ("bgr", C.POINTER(C.c_uint8)), ("d_min", C.c_float),
                ("d_max", C.c_float), ("n_neighbors", C.c_int), ("neighbors", C.POINTER(C.c_uint32))]


class Cloud(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("capacity", C.c_uint64), ("xyz", C.POINTER(C.c_float)),
                ("normal", C.POINTER(C.c_float)), ("bgr", C.POINTER(C.c_uint8)), ("n_views", C.POINTER(C.c_uint32)),
                ("n_depths", C.c_uint64), ("views_capacity", C.c_uint64), ("n_view_entries", C.c_uint64),
                ("view_ids", C.POINTER(C.c_uint32)), ("view_weights", C.POINTER(C.c_float)), ("claim_image", C.c_uint32),
                ("claim_mask", C.POINTER(C.c_uint8))]


_lib = None


def build():
    if not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libhcmvs_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        fp = C.POINTER(C.c_float); u8p = C.POINTER(C.c_uint8)
        L.hcor_default_params.argtypes = [C.POINTER(Params)]
        L.hcor_zigzag_coords.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16)]
        L.hcor_zigzag_coords.restype = C.c_int
        L.hcor_rand_u32.argtypes = [C.c_uint32] * 4
        L.hcor_rand_u32.restype = C.c_uint32
        for n in ("expf", "sinf", "cosf", "acosf"):
            f = getattr(L, "hcor_pm_" + n); f.argtypes = [C.c_float]; f.restype = C.c_float
        L.hcor_pm_atan2f.argtypes = [C.c_float, C.c_float]; L.hcor_pm_atan2f.restype = C.c_float
        L.hcor_bgr2gray_u8.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.hcor_gray_f32_to_u8.argtypes = [fp, C.c_int, C.c_int, u8p]
        L.hcor_gradient_map.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.hcor_median3.argtypes = [fp, C.c_int, C.c_int, fp]
        L.hcor_resize_size.argtypes = [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hcor_resize_gray.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, C.c_int, C.c_int]
        L.hcor_resize_area_up.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.hcor_inside_rule_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
        L.hcor_splat_init.argtypes = [C.POINTER(View), fp, C.c_int, fp, fp, fp, fp]
        L.hcor_fill_patch.argtypes = [C.POINTER(View), u8p, C.POINTER(Params), C.c_int, C.c_int, C.c_int, fp, fp, fp, fp]
        L.hcor_fill_patch.restype = C.c_int
        L.hcor_score_view.argtypes = [C.POINTER(View), C.POINTER(View), u8p, C.POINTER(Params), C.c_int, C.c_int,
                                      C.c_float, fp]
        L.hcor_score_view.restype = C.c_float
        L.hcor_score_pixel.argtypes = [C.POINTER(View), C.POINTER(View), C.c_int, u8p, C.POINTER(Params), C.c_int,
                                       C.c_int, C.c_float, fp]
        L.hcor_score_pixel.restype = C.c_float
        L.hcor_normal2dir.argtypes = [fp, fp, C.c_int]
        L.hcor_dir2normal.argtypes = [fp, fp, C.c_int]
        L.hcor_correct_normal.argtypes = [C.POINTER(View), C.c_int, C.c_int, fp, C.c_int]
        L.hcor_interpolate_pixel.argtypes = [C.POINTER(View), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, fp,
                                             C.c_float, C.c_float]
        L.hcor_interpolate_pixel.restype = C.c_float
        est_args = [C.POINTER(View), C.POINTER(View), C.c_int, u8p, C.POINTER(Params), C.c_float, C.c_float, fp, fp,
                    fp, C.POINTER(C.c_uint64)]
        L.hcor_estimate.argtypes = est_args; L.hcor_estimate.restype = C.c_int
        L.hcor_pass_score.argtypes = est_args
        L.hcor_pass_sweep.argtypes = est_args[:5] + [C.c_int] + est_args[5:]
        L.hcor_pass_end.argtypes = [C.POINTER(Params), C.c_int, C.c_int, fp, fp, fp]
        L.hcor_filter_depthmap.argtypes = [C.POINTER(DepthMap), C.c_uint32, C.POINTER(C.c_uint32), C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_float, fp, fp, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]
        L.hcor_filter_depthmap.restype = C.c_int
        L.hcor_fuse_depthmaps.argtypes = [C.POINTER(DepthMap), C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int,
                                          C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Cloud)]
        L.hcor_fuse_depthmaps.restype = C.c_int
        L.hcor_set_fuse_pixel_order.argtypes = [C.c_int]
        L.hcor_set_fuse_pixel_order.restype = None
        L.hcor_estimate_point_colors.argtypes = [C.POINTER(DepthMap), C.c_int, C.c_uint64, fp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), u8p]
        L.hcor_estimate_point_colors.restype = None
        L.hcor_postfilter.argtypes = [C.POINTER(DepthMap), C.c_int, C.c_uint32, u8p, C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.hcor_postfilter.restype = C.c_int
    return _lib


def fptr(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def default_params(**kw):
    p = Params()
    lib().hcor_default_params(C.byref(p))
    for k, v in kw.items():
        assert hasattr(p, k), k
        setattr(p, k, v)
    return p


def make_view(v):
    """v: dict with gray (h,w) f32, K,R (3,3), C (3).  Keeps references alive on the struct."""
    s = View()
    g = np.ascontiguousarray(v["gray"], np.float32)
    s.width, s.height = g.shape[1], g.shape[0]
    s.gray = fptr(g)
    s._keep = [g]
    if v.get("bgr") is not None:
        b = np.ascontiguousarray(v["bgr"], np.uint8); s.bgr = u8ptr(b); s._keep.append(b)
    s.K[:] = list(np.asarray(v["K"], np.float64).ravel())
    s.R[:] = list(np.asarray(v["R"], np.float64).ravel())
    s.C[:] = list(np.asarray(v["C"], np.float64).ravel())
    return s


def make_view_array(vs):
    arr = (View * len(vs))()
    keep = []
    for i, v in enumerate(vs):
        s = make_view(v)
        keep.append(s._keep)
        C.memmove(C.byref(arr[i]), C.byref(s), C.sizeof(View))
    arr._keep = keep
    return arr


def gradient_map(gray_f32):
    L = lib()
    h, w = gray_f32.shape
    g8 = np.empty((h, w), np.uint8); gra = np.empty((h, w), np.uint8)
    L.hcor_gray_f32_to_u8(fptr(np.ascontiguousarray(gray_f32)), w, h, u8ptr(g8))
    L.hcor_gradient_map(u8ptr(g8), w, h, u8ptr(gra))
    return gra


def resize_gray(gray, scale):
    """ViewData::ScaleImage on an f32 image: returns the resampled image"""
    L = lib()
    g = np.ascontiguousarray(gray, np.float32)
    h, w = g.shape
    dw = C.c_int(); dh = C.c_int()
    L.hcor_resize_size(w, h, C.c_float(scale), C.byref(dw), C.byref(dh))
    out = np.empty((dh.value, dw.value), np.float32)
    L.hcor_resize_gray(fptr(g), w, h, C.c_float(scale), fptr(out), dw.value, dh.value)
    return out


def resize_area_up(src, dw, dh):
    """cv::resize INTER_AREA, enlarging, on an f32 map (h, w) or (h, w, ch)"""
    s = np.ascontiguousarray(src, np.float32)
    ch = 1 if s.ndim == 2 else s.shape[2]
    out = np.empty((dh, dw) if s.ndim == 2 else (dh, dw, ch), np.float32)
    lib().hcor_resize_area_up(fptr(s), s.shape[1], s.shape[0], ch, fptr(out), dw, dh)
    return out


def estimate(views, params, d_min, d_max, depth, normal, gra=None, passes="all", iter_index=0):
    """Runs the oracle on views[0] (ref) vs views[1:].  depth (h,w), normal (h,w,3) are copied.
    Returns depth, normal, conf, evals."""
    L = lib()
    ref = make_view(views[0]); srcs = make_view_array(views[1:])
    h, w = views[0]["gray"].shape
    if gra is None:
        gra = gradient_map(views[0]["gray"])
    d = np.ascontiguousarray(depth, np.float32).copy()
    n = np.ascontiguousarray(normal, np.float32).copy()
    c = np.zeros((h, w), np.float32)
    ev = C.c_uint64(0)
    rc = L.hcor_estimate(C.byref(ref), srcs, len(views) - 1, u8ptr(gra), C.byref(params), d_min, d_max, fptr(d),
                         fptr(n), fptr(c), C.byref(ev))
    assert rc == 0
    return d, n, c, ev.value


def make_depthmaps(maps):
    """maps: list of dicts (K,R,C, depth (h,w), normal (h,w,3) or None, conf, bgr or None, d_min, d_max, neighbors).
    Depth arrays are copied (fusion mutates them); returns (ctypes array, list of the mutable depth copies)."""
    arr = (DepthMap * len(maps))()
    keep = []; depths = []
    for i, m in enumerate(maps):
        d = np.ascontiguousarray(m["depth"], np.float32).copy(); c = np.ascontiguousarray(m["conf"], np.float32)
        arr[i].height, arr[i].width = d.shape
        arr[i].K[:] = list(np.asarray(m["K"], np.float64).ravel()); arr[i].R[:] = list(np.asarray(m["R"], np.float64).ravel())
        arr[i].C[:] = list(np.asarray(m["C"], np.float64).ravel())
        arr[i].depth = fptr(d); arr[i].conf = fptr(c); keep += [d, c]; depths.append(d)
        if m.get("normal") is not None:
            n = np.ascontiguousarray(m["normal"], np.float32); arr[i].normal = fptr(n); keep.append(n)
        if m.get("bgr") is not None:
            b = np.ascontiguousarray(m["bgr"], np.uint8); arr[i].bgr = u8ptr(b); keep.append(b)
        arr[i].d_min = m.get("d_min", 0.1); arr[i].d_max = m.get("d_max", 1e9)
        nb = (C.c_uint32 * max(len(m["neighbors"]), 1))(*m["neighbors"])
        arr[i].neighbors = nb; arr[i].n_neighbors = len(m["neighbors"]); keep.append(nb)
    arr._keep = keep
    return arr, depths


def filter_depthmap(maps, ref_id, neighbor_ids, adjust=True, n_min_views=2, n_min_views_adjust=1, thr=0.01):
    arr, _ = make_depthmaps(maps)
    h, w = maps[ref_id]["depth"].shape
    d = np.empty((h, w), np.float32); c = np.empty((h, w), np.float32)
    ids = (C.c_uint32 * len(neighbor_ids))(*neighbor_ids)
    npr = C.c_uint64(); nd = C.c_uint64()
    ok = lib().hcor_filter_depthmap(arr, ref_id, ids, len(neighbor_ids), int(adjust), n_min_views, n_min_views_adjust, thr,
                                    fptr(d), fptr(c), C.byref(npr), C.byref(nd))
    return ok, d, c, npr.value, nd.value


def fuse_depthmaps(maps, order, capacity, n_min_views_fuse=2, thr=0.01, normal_deg=25.0, depthweight=1.0, normalweight=1.0, pixel_order=0):
    """pixel_order 1: the pixels of an image in the hashed order of hcmvs_set_fuse_order(ctx, 1); the points then come out in visiting
    order (the device writes them in raster order)"""
    arr, depths = make_depthmaps(maps)
    xyz = np.zeros((capacity, 3), np.float32); nrm = np.zeros((capacity, 3), np.float32)
    bgr = np.zeros((capacity, 3), np.uint8); nv = np.zeros(capacity, np.uint32)
    cl = Cloud(); cl.capacity = capacity; cl.xyz = fptr(xyz); cl.normal = fptr(nrm); cl.bgr = u8ptr(bgr)
    cl.n_views = nv.ctypes.data_as(C.POINTER(C.c_uint32))
    vcap = int(sum(int((np.asarray(m["depth"]) != 0).sum()) for m in maps))
    vids = np.zeros(max(vcap, 1), np.uint32); vwts = np.zeros(max(vcap, 1), np.float32)
    cl.views_capacity = vcap; cl.view_ids = vids.ctypes.data_as(C.POINTER(C.c_uint32)); cl.view_weights = fptr(vwts)
    ids = (C.c_uint32 * len(order))(*order)
    lib().hcor_set_fuse_pixel_order(int(pixel_order))
    try:
        rc = lib().hcor_fuse_depthmaps(arr, len(maps), ids, len(order), n_min_views_fuse, thr, normal_deg, depthweight,
                                       normalweight, C.byref(cl))
    finally:
        lib().hcor_set_fuse_pixel_order(0)
    assert rc == 0
    k = cl.n_points
    ne = cl.n_view_entries
    return dict(xyz=xyz[:k], normal=nrm[:k], bgr=bgr[:k], n_views=nv[:k], n_points=k, n_depths=cl.n_depths, depths=depths,
                view_ids=vids[:ne], view_weights=vwts[:ne])


def estimate_point_colors(maps, xyz, n_views, view_ids):
    arr, _ = make_depthmaps(maps)
    x = np.ascontiguousarray(xyz, np.float32); nv = np.ascontiguousarray(n_views, np.uint32); vi = np.ascontiguousarray(view_ids, np.uint32)
    out = np.zeros((len(x), 3), np.uint8)
    lib().hcor_estimate_point_colors(arr, len(maps), len(x), fptr(x), nv.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     vi.ctypes.data_as(C.POINTER(C.c_uint32)), u8ptr(out))
    return out


def postfilter(maps, vid, gra, order, mode=ARITH_DEVICE, n_min_views_fuse=2, thr=0.01, normal_deg=25.0, gap=7):
    """RemoveSmallSegments (fork) + GapInterpolation on image vid; maps are copied.  Returns (maps' depth copies, normal, conf of vid, n_filled)"""
    maps = [dict(m) for m in maps]
    maps[vid]["normal"] = np.ascontiguousarray(maps[vid]["normal"], np.float32).copy()
    maps[vid]["conf"] = np.ascontiguousarray(maps[vid]["conf"], np.float32).copy()
    arr, depths = make_depthmaps(maps)
    arr[vid].normal = fptr(maps[vid]["normal"]); arr[vid].conf = fptr(maps[vid]["conf"])
    ids = (C.c_uint32 * len(order))(*order)
    nf = C.c_uint64()
    g = np.ascontiguousarray(gra, np.uint8)
    rc = lib().hcor_postfilter(arr, len(maps), vid, u8ptr(g), ids, len(order), n_min_views_fuse, thr, normal_deg, gap, mode,
                               C.byref(nf))
    assert rc == 0
    return depths, maps[vid]["normal"], maps[vid]["conf"], nf.value
