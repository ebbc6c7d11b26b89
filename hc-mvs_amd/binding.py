"""ctypes binding of libhcmvs_hip.so (the C-ABI of include/hcmvs_hip.h).

Mirrors the reference's per-image interface (SceneDensify.h:63-70 DepthMapsData::EstimateDepthMap /
FilterDepthMap / FuseDepthMaps) for tests and bench.py.  There is no CPU fallback: loading fails loudly
when the library is missing, and creating a context fails when no gfx950 device is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhcmvs_hip.so")

OK, ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_TIMEOUT, ERR_CAPACITY = range(6)


class Params(C.Structure):
    _fields_ = [("adapthalfwin", C.c_int32), ("n_estimation_iters", C.c_int32), ("it_external", C.c_int32),
                ("n_external_iters", C.c_int32), ("propagate_halfwin", C.c_int32), ("propagate_step", C.c_int32),
                ("n_random_iters", C.c_int32), ("ncc_threshold_keep", C.c_float), ("random_depth_ratio", C.c_float),
                ("random_angle1_deg", C.c_float), ("random_angle2_deg", C.c_float),
                ("random_smooth_depth", C.c_float), ("random_smooth_normal_deg", C.c_float),
                ("random_smooth_bonus", C.c_float), ("photometric_flow", C.c_float), ("seed", C.c_uint32),
                ("median_blur", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("evals", C.c_uint64), ("evals_issued", C.c_uint64), ("tap_evals", C.c_uint64), ("ms_score", C.c_float), ("ms_sweeps", C.c_float),
                ("ms_sweep_avg", C.c_float), ("ms_end", C.c_float), ("ms_total", C.c_float),
                ("n_sweeps", C.c_int32), ("n_sweep_launches", C.c_int32)]


class BatchItem(C.Structure):
    _fields_ = [("ref_id", C.c_uint32), ("src_ids", C.POINTER(C.c_uint32)), ("n_src", C.c_int32),
                ("seed_offset", C.c_uint32), ("d_min", C.c_float), ("d_max", C.c_float), ("d_depth", C.c_void_p),
                ("d_normal", C.c_void_p), ("d_conf", C.c_void_p), ("d_hint_depth", C.c_void_p), ("d_hint_normal", C.c_void_p)]


class Cloud(C.Structure):
    _fields_ = [("capacity", C.c_uint64), ("xyz", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)), ("bgr", C.POINTER(C.c_uint8)),
                ("n_views", C.POINTER(C.c_uint32)), ("views_capacity", C.c_uint64), ("view_ids", C.POINTER(C.c_uint32)),
                ("view_weights", C.POINTER(C.c_float)), ("n_points", C.c_uint64), ("n_depths", C.c_uint64), ("n_view_entries", C.c_uint64)]


class HcmvsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hcmvs error %d: %s" % (code, msg))
        self.code = code


_lib = None

# every symbol include/hcmvs_hip.h declares
SYMBOLS = ["hcmvs_default_params", "hcmvs_create", "hcmvs_destroy", "hcmvs_last_error", "hcmvs_set_stream",
           "hcmvs_synchronize", "hcmvs_upload_view", "hcmvs_set_view_device", "hcmvs_release_view", "hcmvs_rescale_view", "hcmvs_get_view_info",
           "hcmvs_get_view_gray",
           "hcmvs_get_gradient_map", "hcmvs_estimate", "hcmvs_estimate_device", "hcmvs_estimate_batch_device", "hcmvs_get_stats",
           "hcmvs_splat_init", "hcmvs_splat_points", "hcmvs_triangulate_init", "hcmvs_triangulate_points", "hcmvs_set_depthmap", "hcmvs_set_depthmap_device", "hcmvs_get_depthmap",
           "hcmvs_set_neighbors", "hcmvs_filter", "hcmvs_set_fuse_order", "hcmvs_fuse", "hcmvs_fuse_cloud", "hcmvs_estimate_point_colors",
           "hcmvs_estimate_point_normals", "hcmvs_postfilter", "hcmvs_postfilter_sequence", "hcmvs_resize_area_up"]


def triangulate_points(w, h, K, R, Cc, points_xyz, avg_depth=0.0, add_corners=True):
    """hcmvs_triangulate_points: the triangulation initialisation without a context (pure host code)"""
    pts = np.ascontiguousarray(points_xyz, np.float32)
    depth = np.zeros((h, w), np.float32); normal = np.zeros((h, w, 3), np.float32)
    dmin = C.c_float(); dmax = C.c_float()
    Ka, Kp = _d(K); Ra, Rp = _d(R); Ca, Cp = _d(Cc)
    rc = lib().hcmvs_triangulate_points(w, h, Kp, Rp, Cp, _f(pts), len(pts), C.c_float(avg_depth), int(add_corners), _f(depth),
                                        _f(normal), C.byref(dmin), C.byref(dmax))
    if rc != 0:
        raise HcmvsError(rc, "triangulate_points failed")
    return depth, normal, dmin.value, dmax.value


def resize_area_up(src, dst_w, dst_h):
    """hcmvs_resize_area_up: cv::resize INTER_AREA enlarging an f32 map (h, w) or (h, w, channels); pure host code"""
    s = np.ascontiguousarray(src, np.float32)
    ch = 1 if s.ndim == 2 else s.shape[2]
    out = np.empty((dst_h, dst_w) if s.ndim == 2 else (dst_h, dst_w, ch), np.float32)
    rc = lib().hcmvs_resize_area_up(_f(s), s.shape[1], s.shape[0], ch, _f(out), dst_w, dst_h)
    if rc != 0:
        raise HcmvsError(rc, "resize_area_up failed")
    return out


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libhcmvs_hip.so is not built (run __graft_entry__.build() or make -C hc-mvs_amd/csrc); "
                              "there is no CPU fallback")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process cannot both
        # open the device.  Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to the
        # copy that is already mapped, so the process holds exactly one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, fp, u8p, dp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        u32p = C.POINTER(C.c_uint32)
        L.hcmvs_default_params.argtypes = [C.POINTER(Params)]
        L.hcmvs_default_params.restype = None
        L.hcmvs_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.hcmvs_destroy.argtypes = [vp]
        L.hcmvs_destroy.restype = None
        L.hcmvs_last_error.argtypes = [vp]
        L.hcmvs_last_error.restype = C.c_char_p
        L.hcmvs_set_stream.argtypes = [vp, vp]
        L.hcmvs_synchronize.argtypes = [vp]
        L.hcmvs_upload_view.argtypes = [vp, C.c_uint32, C.c_int32, C.c_int32, fp, u8p, dp, dp, dp]
        L.hcmvs_set_view_device.argtypes = [vp, C.c_uint32, C.c_int32, C.c_int32, vp, vp, dp, dp, dp]
        L.hcmvs_release_view.argtypes = [vp, C.c_uint32]
        L.hcmvs_rescale_view.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float]
        L.hcmvs_get_view_info.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp]
        L.hcmvs_get_view_gray.argtypes = [vp, C.c_uint32, fp]
        L.hcmvs_get_gradient_map.argtypes = [vp, C.c_uint32, u8p]
        L.hcmvs_estimate.argtypes = [vp, C.c_uint32, u32p, C.c_int32, C.POINTER(Params), C.c_float, C.c_float, fp, fp, fp]
        L.hcmvs_estimate_device.argtypes = [vp, C.c_uint32, u32p, C.c_int32, C.POINTER(Params), C.c_float, C.c_float,
                                            vp, vp, vp]
        L.hcmvs_estimate_batch_device.argtypes = [vp, C.POINTER(BatchItem), C.c_int32, C.POINTER(Params)]
        L.hcmvs_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.hcmvs_splat_init.argtypes = [vp, C.c_uint32, fp, C.c_int32, fp, fp, fp, fp]
        L.hcmvs_splat_points.argtypes = [C.c_int32, C.c_int32, dp, dp, dp, fp, C.c_int32, fp, fp, fp, fp]
        L.hcmvs_triangulate_init.argtypes = [vp, C.c_uint32, fp, C.c_int32, C.c_float, C.c_int32, fp, fp, fp, fp]
        dp = C.POINTER(C.c_double)
        L.hcmvs_triangulate_points.argtypes = [C.c_int32, C.c_int32, dp, dp, dp, fp, C.c_int32, C.c_float, C.c_int32, fp, fp, fp, fp]
        u64p = C.POINTER(C.c_uint64)
        L.hcmvs_set_depthmap.argtypes = [vp, C.c_uint32, fp, fp, fp, C.c_float, C.c_float]
        L.hcmvs_set_depthmap_device.argtypes = [vp, C.c_uint32, vp, vp, vp, C.c_float, C.c_float]
        L.hcmvs_get_depthmap.argtypes = [vp, C.c_uint32, fp, fp, fp]
        L.hcmvs_set_neighbors.argtypes = [vp, C.c_uint32, u32p, C.c_int32]
        L.hcmvs_set_fuse_order.argtypes = [vp, C.c_int32]
        L.hcmvs_filter.argtypes = [vp, C.c_uint32, u32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, fp, fp,
                                   u64p, u64p]
        L.hcmvs_fuse.argtypes = [vp, u32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint64,
                                 fp, fp, u8p, u32p, u64p, u64p]
        L.hcmvs_fuse_cloud.argtypes = [vp, u32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Cloud)]
        L.hcmvs_postfilter.argtypes = [vp, C.c_uint32, u32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, u64p]
        L.hcmvs_postfilter_sequence.argtypes = [vp, u32p, C.c_int32, u32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, u64p]
        L.hcmvs_estimate_point_colors.argtypes = [vp, C.c_uint64, fp, u32p, u32p, u8p]
        L.hcmvs_estimate_point_normals.argtypes = [vp, C.c_uint64, fp, u32p, u32p, C.c_int32, fp]
        L.hcmvs_resize_area_up.argtypes = [fp, C.c_int32, C.c_int32, C.c_int32, fp, C.c_int32, C.c_int32]
        _lib = L
    return _lib


def default_params(**kw):
    p = Params()
    lib().hcmvs_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _f(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    a = np.ascontiguousarray(a, np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """One device context (hcmvs_ctx).  Host-buffer methods mirror DepthMapsData's per-image calls."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().hcmvs_create(device, C.byref(self._h))
        if rc != OK:
            raise HcmvsError(rc, "hcmvs_create failed (no usable gfx950 device?)")
        self.shapes = {}

    def close(self):
        if self._h:
            lib().hcmvs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise HcmvsError(rc, lib().hcmvs_last_error(self._h).decode())

    def set_stream(self, stream_ptr):
        self._chk(lib().hcmvs_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._chk(lib().hcmvs_synchronize(self._h))

    def upload_view(self, vid, gray, K, R, Cc, bgr=None):
        """gray None (with a colour image): a fuse-only view -- camera, colours, gradient map, no estimate"""
        Ka, Kp = _d(K); Ra, Rp = _d(R); Ca, Cp = _d(Cc)
        bp = None
        if bgr is not None:
            bgr = np.ascontiguousarray(bgr, np.uint8)
            bp = bgr.ctypes.data_as(C.POINTER(C.c_uint8))
        if gray is not None:
            gray = np.ascontiguousarray(gray, np.float32)
            h, w = gray.shape
        else:
            h, w = bgr.shape[:2]
        self._chk(lib().hcmvs_upload_view(self._h, vid, w, h, None if gray is None else _f(gray), bp, Kp, Rp, Cp))
        self.shapes[vid] = (h, w)

    def set_view_device(self, vid, w, h, d_gray_ptr, K, R, Cc, d_bgr_ptr=None):
        Ka, Kp = _d(K); Ra, Rp = _d(R); Ca, Cp = _d(Cc)
        self._chk(lib().hcmvs_set_view_device(self._h, vid, w, h, C.c_void_p(d_gray_ptr),
                                              C.c_void_p(d_bgr_ptr) if d_bgr_ptr else None, Kp, Rp, Cp))
        self.shapes[vid] = (h, w)

    def rescale_view(self, src_id, dst_id, scale):
        """DepthData::ViewData::ScaleImage: view dst_id = view src_id resampled by scale; returns (gray, K) of the new view"""
        self._chk(lib().hcmvs_rescale_view(self._h, src_id, dst_id, C.c_float(scale)))
        w = C.c_int32(); h = C.c_int32(); K = (C.c_double * 9)()
        self._chk(lib().hcmvs_get_view_info(self._h, dst_id, C.byref(w), C.byref(h), K))
        self.shapes[dst_id] = (h.value, w.value)
        g = np.empty((h.value, w.value), np.float32)
        self._chk(lib().hcmvs_get_view_gray(self._h, dst_id, _f(g)))
        return g, np.array(list(K), np.float64).reshape(3, 3)

    def release_view(self, vid):
        self._chk(lib().hcmvs_release_view(self._h, vid))
        self.shapes.pop(vid, None)

    def gradient_map(self, vid):
        h, w = self.shapes[vid]
        out = np.empty((h, w), np.uint8)
        self._chk(lib().hcmvs_get_gradient_map(self._h, vid, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def splat_init(self, vid, points_xyz):
        h, w = self.shapes[vid]
        pts = np.ascontiguousarray(points_xyz, np.float32)
        depth = np.zeros((h, w), np.float32); normal = np.zeros((h, w, 3), np.float32)
        dmin = C.c_float(); dmax = C.c_float()
        self._chk(lib().hcmvs_splat_init(self._h, vid, _f(pts), len(pts), _f(depth), _f(normal), C.byref(dmin),
                                         C.byref(dmax)))
        return depth, normal, dmin.value, dmax.value

    def triangulate_init(self, vid, points_xyz, avg_depth=0.0, add_corners=True):
        """TriangulatePoints2DepthMap as InitDepthMap uses it: returns (depth, normal, d_min, d_max)"""
        h, w = self.shapes[vid]
        pts = np.ascontiguousarray(points_xyz, np.float32)
        depth = np.zeros((h, w), np.float32); normal = np.zeros((h, w, 3), np.float32)
        dmin = C.c_float(); dmax = C.c_float()
        self._chk(lib().hcmvs_triangulate_init(self._h, vid, _f(pts), len(pts), C.c_float(avg_depth), int(add_corners), _f(depth),
                                               _f(normal), C.byref(dmin), C.byref(dmax)))
        return depth, normal, dmin.value, dmax.value

    def estimate(self, ref_id, src_ids, params, d_min, d_max, depth, normal, conf=None):
        """EstimateDepthMap on host maps (copied); returns (depth, normal, conf)."""
        h, w = self.shapes[ref_id]
        d = np.ascontiguousarray(depth, np.float32).copy()
        n = np.ascontiguousarray(normal, np.float32).copy()
        c = np.zeros((h, w), np.float32) if conf is None else np.ascontiguousarray(conf, np.float32).copy()
        assert d.shape == (h, w) and n.shape == (h, w, 3)
        ids = (C.c_uint32 * len(src_ids))(*src_ids)
        self._chk(lib().hcmvs_estimate(self._h, ref_id, ids, len(src_ids), C.byref(params), d_min, d_max, _f(d),
                                       _f(n), _f(c)))
        return d, n, c

    def estimate_device(self, ref_id, src_ids, params, d_min, d_max, d_depth_ptr, d_normal_ptr, d_conf_ptr):
        ids = (C.c_uint32 * len(src_ids))(*src_ids)
        self._chk(lib().hcmvs_estimate_device(self._h, ref_id, ids, len(src_ids), C.byref(params), d_min, d_max,
                                              C.c_void_p(d_depth_ptr), C.c_void_p(d_normal_ptr),
                                              C.c_void_p(d_conf_ptr)))

    def estimate_batch_device(self, items, params):
        """items: list of dicts (ref_id, src_ids, d_min, d_max, d_depth, d_normal, d_conf [device pointers], seed_offset)"""
        arr = (BatchItem * len(items))()
        keep = []
        for i, it in enumerate(items):
            ids = (C.c_uint32 * len(it["src_ids"]))(*it["src_ids"]); keep.append(ids)
            arr[i].ref_id = it["ref_id"]; arr[i].src_ids = ids; arr[i].n_src = len(it["src_ids"])
            arr[i].seed_offset = it.get("seed_offset", 0); arr[i].d_min = it["d_min"]; arr[i].d_max = it["d_max"]
            arr[i].d_depth = it["d_depth"]; arr[i].d_normal = it["d_normal"]; arr[i].d_conf = it["d_conf"]
            arr[i].d_hint_depth = it.get("d_hint_depth"); arr[i].d_hint_normal = it.get("d_hint_normal")
        self._chk(lib().hcmvs_estimate_batch_device(self._h, arr, len(items), C.byref(params)))

    def stats(self):
        s = Stats()
        self._chk(lib().hcmvs_get_stats(self._h, C.byref(s)))
        return s

    # ---- filter / fuse (SceneDensify.h:69-70) -----------------------------------------------------------

    def set_depthmap(self, vid, depth, normal, conf, d_min, d_max):
        d = np.ascontiguousarray(depth, np.float32); c = np.ascontiguousarray(conf, np.float32)
        n = None if normal is None else np.ascontiguousarray(normal, np.float32)
        self._chk(lib().hcmvs_set_depthmap(self._h, vid, _f(d), None if n is None else _f(n), _f(c), d_min, d_max))

    def set_depthmap_device(self, vid, d_depth_ptr, d_normal_ptr, d_conf_ptr, d_min, d_max):
        """maps that already live in device memory (caller-owned; fusion mutates the depth map, SceneDensify.cpp:3447-3449)"""
        self._chk(lib().hcmvs_set_depthmap_device(self._h, vid, C.c_void_p(d_depth_ptr), C.c_void_p(d_normal_ptr) if d_normal_ptr else None,
                                                  C.c_void_p(d_conf_ptr), d_min, d_max))

    def get_depthmap(self, vid, with_normal=False):
        h, w = self.shapes[vid]
        d = np.empty((h, w), np.float32); c = np.empty((h, w), np.float32)
        n = np.empty((h, w, 3), np.float32) if with_normal else None
        self._chk(lib().hcmvs_get_depthmap(self._h, vid, _f(d), None if n is None else _f(n), _f(c)))
        return (d, n, c) if with_normal else (d, c)

    def set_neighbors(self, vid, ids):
        arr = (C.c_uint32 * max(len(ids), 1))(*ids)
        self._chk(lib().hcmvs_set_neighbors(self._h, vid, arr, len(ids)))

    def set_fuse_order(self, mode):
        """0: raster order (bit-exact with the reference's sequential fusion), 1: hashed order (few rounds)"""
        self._chk(lib().hcmvs_set_fuse_order(self._h, int(mode)))

    def filter(self, ref_id, neighbor_ids, adjust=True, n_min_views=2, n_min_views_adjust=1, depth_diff_threshold=0.01):
        """FilterDepthMap: returns (new_depth, new_conf, n_processed, n_discarded)"""
        h, w = self.shapes[ref_id]
        d = np.empty((h, w), np.float32); c = np.empty((h, w), np.float32)
        ids = (C.c_uint32 * len(neighbor_ids))(*neighbor_ids)
        npr = C.c_uint64(); nd = C.c_uint64()
        self._chk(lib().hcmvs_filter(self._h, ref_id, ids, len(neighbor_ids), int(adjust), n_min_views, n_min_views_adjust,
                                     depth_diff_threshold, _f(d), _f(c), C.byref(npr), C.byref(nd)))
        return d, c, npr.value, nd.value

    def fuse(self, order, capacity, n_min_views_fuse=2, depth_diff_threshold=0.01, normal_diff_deg=25.0, depthweight=1.0,
             normalweight=1.0, with_normals=True, with_colors=True):
        """FuseDepthMaps: returns dict(xyz, normal, bgr, n_views, n_points, n_depths)"""
        xyz = np.empty((capacity, 3), np.float32)  # only the first n_points rows are written and returned
        nrm = np.empty((capacity, 3), np.float32) if with_normals else None
        bgr = np.empty((capacity, 3), np.uint8) if with_colors else None
        nv = np.empty(capacity, np.uint32)
        ids = (C.c_uint32 * len(order))(*order)
        npts = C.c_uint64(); nd = C.c_uint64()
        self._chk(lib().hcmvs_fuse(self._h, ids, len(order), n_min_views_fuse, depth_diff_threshold, normal_diff_deg,
                                   depthweight, normalweight, capacity, _f(xyz), None if nrm is None else _f(nrm),
                                   None if bgr is None else bgr.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   nv.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(npts), C.byref(nd)))
        k = npts.value
        return dict(xyz=xyz[:k], normal=None if nrm is None else nrm[:k], bgr=None if bgr is None else bgr[:k],
                    n_views=nv[:k], n_points=k, n_depths=nd.value)

    def fuse_count(self, order, n_min_views_fuse=2, depth_diff_threshold=0.01, normal_diff_deg=25.0, depthweight=1.0, normalweight=1.0):
        """hcmvs_fuse_cloud without buffers: the fusion runs (with its side effect, the invalidated depths) and only counts.  A fusion
        repeated on the maps it has left makes the same decisions, so buffers of exactly (n_points, n_view_entries) hold the cloud of the
        fusion that follows.  Returns (n_points, n_depths, n_view_entries)."""
        cl = Cloud()
        ids = (C.c_uint32 * len(order))(*order)
        self._chk(lib().hcmvs_fuse_cloud(self._h, ids, len(order), n_min_views_fuse, depth_diff_threshold, normal_diff_deg, depthweight, normalweight,
                                         C.byref(cl)))
        return cl.n_points, cl.n_depths, cl.n_view_entries

    def fuse_cloud(self, order, capacity, views_capacity, n_min_views_fuse=2, depth_diff_threshold=0.01, normal_diff_deg=25.0, depthweight=1.0,
                   normalweight=1.0):
        """hcmvs_fuse_cloud: the complete PointCloud (points, view lists + weights as CSR, colours, normals)"""
        xyz = np.empty((capacity, 3), np.float32); nrm = np.empty((capacity, 3), np.float32); bgr = np.empty((capacity, 3), np.uint8)
        nv = np.empty(capacity, np.uint32); vids = np.empty(max(views_capacity, 1), np.uint32); vwts = np.empty(max(views_capacity, 1), np.float32)
        cl = Cloud()
        cl.capacity = capacity; cl.xyz = _f(xyz); cl.normal = _f(nrm); cl.bgr = bgr.ctypes.data_as(C.POINTER(C.c_uint8))
        cl.n_views = nv.ctypes.data_as(C.POINTER(C.c_uint32)); cl.views_capacity = views_capacity
        cl.view_ids = vids.ctypes.data_as(C.POINTER(C.c_uint32)); cl.view_weights = _f(vwts)
        ids = (C.c_uint32 * len(order))(*order)
        self._chk(lib().hcmvs_fuse_cloud(self._h, ids, len(order), n_min_views_fuse, depth_diff_threshold, normal_diff_deg, depthweight, normalweight,
                                         C.byref(cl)))
        k, ne = cl.n_points, cl.n_view_entries
        return dict(xyz=xyz[:k], normal=nrm[:k], bgr=bgr[:k], n_views=nv[:k], n_points=k, n_depths=cl.n_depths, view_ids=vids[:ne],
                    view_weights=vwts[:ne])

    def estimate_point_colors(self, xyz, n_views, view_ids):
        x = np.ascontiguousarray(xyz, np.float32); nv = np.ascontiguousarray(n_views, np.uint32); vi = np.ascontiguousarray(view_ids, np.uint32)
        out = np.empty((len(x), 3), np.uint8)
        self._chk(lib().hcmvs_estimate_point_colors(self._h, len(x), _f(x), nv.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    vi.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def estimate_point_normals(self, xyz, n_views, view_ids, n_neighbors=16):
        x = np.ascontiguousarray(xyz, np.float32); nv = np.ascontiguousarray(n_views, np.uint32); vi = np.ascontiguousarray(view_ids, np.uint32)
        out = np.empty((len(x), 3), np.float32)
        self._chk(lib().hcmvs_estimate_point_normals(self._h, len(x), _f(x), nv.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     vi.ctypes.data_as(C.POINTER(C.c_uint32)), n_neighbors, _f(out)))
        return out

    def postfilter(self, vid, order, n_min_views_fuse=2, depth_diff_threshold=0.01, normal_diff_deg=25.0, gap_size=7):
        """RemoveSmallSegments (fork version) + GapInterpolation on the registered device maps of view vid; returns pixels filled"""
        ids = (C.c_uint32 * len(order))(*order)
        nf = C.c_uint64()
        self._chk(lib().hcmvs_postfilter(self._h, vid, ids, len(order), n_min_views_fuse, depth_diff_threshold, normal_diff_deg, gap_size, C.byref(nf)))
        return nf.value

    def postfilter_sequence(self, vids, order, n_min_views_fuse=2, depth_diff_threshold=0.01, normal_diff_deg=25.0, gap_size=7):
        """the post-filters for the images vids one after the other (= postfilter(v) for v in vids); returns the pixels filled in total"""
        ids = (C.c_uint32 * len(vids))(*vids)
        ords = (C.c_uint32 * len(order))(*order)
        nf = C.c_uint64()
        self._chk(lib().hcmvs_postfilter_sequence(self._h, ids, len(vids), ords, len(order), n_min_views_fuse, depth_diff_threshold, normal_diff_deg,
                                                  gap_size, C.byref(nf)))
        return nf.value
