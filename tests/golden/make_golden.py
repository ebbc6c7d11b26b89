"""Generates tests/golden/estimate_96x80_v3.npz: a small synthetic scene plus the oracle's maps for it
(reference arithmetic, zig-zag order, one thread).  Run from the repo root: python tests/golden/make_golden.py"""
import ctypes as C, importlib, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import oracle_lib as O
synth = importlib.import_module("hc-mvs_amd.synth")
w, h, f, V, seed = 96, 80, 90.0, 3, 2
views = synth.make_views(w, h, f, V, seed=seed); pts = synth.sparse_points(views, 70)
L = O.lib(); ref = O.make_view(views[0])
d0 = np.zeros((h, w), np.float32); n0 = np.zeros((h, w, 3), np.float32); dmin = C.c_float(); dmax = C.c_float()
L.hcor_splat_init(C.byref(ref), O.fptr(pts), len(pts), O.fptr(d0), O.fptr(n0), C.byref(dmin), C.byref(dmax))
p = O.default_params(adapthalfwin=6, n_estimation_iters=3, seed=4321, order=O.ORDER_ZIGZAG, n_threads=1)
d, n, c, ev = O.estimate(views, p, dmin.value, dmax.value, d0, n0)
np.savez_compressed(os.path.join(HERE, "estimate_96x80_v3.npz"),
                    gray=np.stack([v["gray"] for v in views]), K=np.stack([v["K"] for v in views]),
                    R=np.stack([v["R"] for v in views]), C=np.stack([v["C"] for v in views]),
                    d0=d0, n0=n0, dmin=dmin.value, dmax=dmax.value, seed=4321, depth=d, normal=n, conf=c, evals=ev)
print("written", ev, (d > 0).mean())

# the same scene in the DEVICE association (the IEEE operation sequence of the gfx950 kernels; it changes whenever the
# kernels' association changes -- regenerate then): the bits the GPU must reproduce through the C-ABI
pd = O.default_params(adapthalfwin=6, n_estimation_iters=3, seed=4321, arith_mode=O.ARITH_DEVICE, order=O.ORDER_ROWS, n_threads=4)
dd, dn, dc, dev_ = O.estimate(views, pd, dmin.value, dmax.value, d0, n0)
np.savez_compressed(os.path.join(HERE, "estimate_96x80_v3_device.npz"), depth=dd, normal=dn, conf=dc, evals=dev_)
print("written (device association)", dev_, (dd > 0).mean())
