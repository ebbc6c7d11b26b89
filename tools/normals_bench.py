"""diagnostic (not a test): time hcmvs_estimate_point_normals on a synthetic surface cloud of N points"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
binding = importlib.import_module("hc-mvs_amd.binding")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
rng = np.random.RandomState(1)
u, v = rng.uniform(-1, 1, N), rng.uniform(-1, 1, N)
z = 5 + 0.3 * np.sin(3 * u) * np.cos(2 * v)
xyz = np.stack([u * 2, v * 1.5, z], -1).astype(np.float32)
xyz[: N // 100] += rng.normal(0, 0.5, (N // 100, 3)).astype(np.float32)          # 1 % outliers off the surface
ctx = binding.Context(0)
g = np.zeros((16, 16), np.float32)
ctx.upload_view(0, g, np.eye(3), np.eye(3), np.array([0.0, 0.0, 0.0]))
nv = np.ones(N, np.uint32); vid = np.zeros(N, np.uint32)
for rep in range(2):
    t0 = time.perf_counter()
    n = ctx.estimate_point_normals(xyz, nv, vid, 16)
    dt = time.perf_counter() - t0
print("normals of %d points in %.3f s (%.1f Mpoints/s); mean |n.z| %.3f, facing the camera %.3f" % (N, dt, N / dt / 1e6, np.abs(n[:, 2]).mean(), ((n * -xyz).sum(1) > 0).mean()))
ctx.close()
