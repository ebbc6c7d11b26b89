"""ad-hoc (not a test): BASELINE.json configs[2] -- a synthetic N-image 1080p scene through the stand-alone driver
(scene.mvs + PPM images -> view selection -> triangulated init -> batched EstimateDepthMap -> DR depth maps -> fuse ->
.ply/.mvs).  Prints the driver's wall time and the per-stage rates it logs.
  python tests/scene_bench.py [n_images=64] [w=1920] [h=1080] [sweeps=8]"""
import importlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
synth = importlib.import_module("hc-mvs_amd.synth")
mvsio = importlib.import_module("hc-mvs_amd.mvsio")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
SWEEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="hcmvs_scene_")
t0 = time.time()
f = 1600.0 * W / 1920
px = 10.0 / f
scene = synth.Scene(3, min_wavelength=3.5 * px, max_wavelength=150 * px)
K = np.array([[f, 0, (W - 1) / 2.0], [0, f, (H - 1) / 2.0], [0, 0, 1]], np.float64)
target = np.array([0.0, 0.0, scene.depth0])
views = []
rng = np.random.RandomState(11)
for i in range(N):  # two rings of cameras looking at the object
    ring = i % 2
    ang = 2 * np.pi * (i // 2) / (N // 2)
    rad = scene.depth0 * (0.10 + 0.06 * ring)
    C = np.array([rad * np.cos(ang), rad * np.sin(ang) * 0.7, 0.01 * rng.uniform(-1, 1)])
    R = synth.look_at(C, target)
    gray, depth, normal = scene.render(K, R, C, W, H)
    views.append(dict(K=K, R=R, C=C, gray=gray, depth=depth))
poses, images = [], []
for i, v in enumerate(views):
    g8 = np.clip(np.rint(v["gray"] * 255), 0, 255).astype(np.uint8)
    mvsio.write_pgm(os.path.join(tmp, "view%03d.pgm" % i), g8)
    poses.append(dict(R=v["R"], C=v["C"]))
    images.append(dict(name="view%03d.pgm" % i, platformID=0, cameraID=0, poseID=i, ID=i))
cams = [dict(name="cam", width=W, height=H, K=K, R=np.eye(3), C=np.zeros(3))]
# sparse points: sampled on the surface seen by every 4th view, with exact visibility lists
verts = []
for i in range(0, N, 4):
    v = views[i]
    xs = rng.randint(10, W - 10, 400); ys = rng.randint(10, H - 10, 400)
    z = v["depth"][ys, xs].astype(np.float64)
    Xc = np.stack([(xs - K[0, 2]) * z / f, (ys - K[1, 2]) * z / f, z], -1)
    Xw = Xc @ v["R"] + v["C"]
    for X in Xw:
        seen = []
        for j, u in enumerate(views):
            p = u["R"] @ (X - u["C"])
            if p[2] <= 0:
                continue
            x, y = f * p[0] / p[2] + K[0, 2], f * p[1] / p[2] + K[1, 2]
            if 2 <= x < W - 2 and 2 <= y < H - 2 and abs(u["depth"][int(round(y)), int(round(x))] - p[2]) < 0.01 * p[2]:
                seen.append((j, 1.0))
        if len(seen) >= 2:
            verts.append(dict(X=X.astype(np.float32), views=seen))
mvsio.write_mvs(os.path.join(tmp, "scene.mvs"), [dict(name="rig", cameras=cams, poses=poses)], images, verts)
print("scene: %d images %dx%d, %d sparse points, generated in %.1f s -> %s" % (N, W, H, len(verts), time.time() - t0, tmp), flush=True)
exe = os.path.join(ROOT, "hc-mvs_amd", "DensifyPointCloud")
t1 = time.time()
r = subprocess.run([exe, "-i", os.path.join(tmp, "scene.mvs"), "-o", os.path.join(tmp, "dense.mvs"), "--resolution-level", "0",
                    "--number-views", "8", "--n-EstimationIters", str(SWEEPS), "--n-EstimationIters-external", "1", "--batch", os.environ.get("BATCH", "32"), "--resume", "0", "-v", "3", "--fuse-order", os.environ.get("FUSE_ORDER", "1")],
                   capture_output=True, text=True)
dt = time.time() - t1
lines = r.stdout.split('\n')
batch_ms = sorted(set(l.split('batch ')[1] for l in lines if 'batch ' in l))
print('\n'.join(l for l in lines if 'batch ' not in l and 'paired with' not in l)[-1500:]); print('batches:', batch_ms); print(r.stderr[-800:])
nsrc = [l.split('using')[1].split('images')[0].strip() for l in lines if 'estimated using' in l]
import collections; print('source views per image:', dict(collections.Counter(nsrc)))
print("driver wall time %.2f s for %d images (%.2f Mpix/s end to end incl. image loading, init, depth-map files, fuse, outputs)" % (
    dt, N, N * W * H / dt / 1e6))
acc = []
for i in range(0, N, 8):
    dm = mvsio.read_dmap(os.path.join(tmp, "depth%04d.dmap" % i))
    m = dm["depth"] > 0
    gt = views[i]["depth"]
    acc.append((m.mean(), (np.abs(dm["depth"] - gt)[m] / gt[m] < 0.01).mean()))
print("valid fraction %.3f, within 1%% of ground truth %.3f (sampled images)" % tuple(np.mean(acc, 0)))
