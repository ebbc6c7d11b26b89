"""diagnostic (not a test): the reference authors' schedule (data/frame_main/resize3/run.py:35-78 -- 10 source views, 4 outer x 3 inner
sweeps, cross pattern 5 / 4, photometric_flow 0.26, --n-nOptimize 1 = post-filters after outer iterations 1 and 2) on the BASELINE
configs[2] scene (N x 1080p on two rings) through the stand-alone driver; prints the driver's per-phase lines.
  python tools/authors_schedule.py [n_images=64] [w=1920] [h=1080] [extra driver flags ...]"""
import importlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
synth = importlib.import_module("hc-mvs_amd.synth")
import scene_files as SF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
extra = sys.argv[4:]
scene_only = "--scene-only" in extra
extra = [e for e in extra if e != "--scene-only"]
f = 1600.0 * W / 1920
px = 10.0 / f
scene = synth.Scene(3, min_wavelength=3.5 * px, max_wavelength=150 * px)
K = np.array([[f, 0, (W - 1) / 2.0], [0, f, (H - 1) / 2.0], [0, 0, 1]], np.float64)
target = np.array([0.0, 0.0, scene.depth0])
rng = np.random.RandomState(11)
poses = []
for i in range(N):
    ring = i % 2
    ang = 2 * np.pi * (i // 2) / (N // 2)
    rad = scene.depth0 * (0.10 + 0.06 * ring)
    Cc = np.array([rad * np.cos(ang), rad * np.sin(ang) * 0.7, 0.01 * rng.uniform(-1, 1)])
    poses.append((synth.look_at(Cc, target), Cc))
t0 = time.time()
views = SF.render_views(scene, K, poses, W, H, threads=12)
verts = SF.sparse_vertices(views, 400, every=4, seed=11)
tmp = tempfile.mkdtemp(prefix="hcmvs_authors_")
path = SF.write_scene(tmp, views, verts)
print("scene: %d images %dx%d, %d sparse points (%.1f s)" % (N, W, H, len(verts), time.time() - t0), flush=True)
if scene_only:   # for tools/prof_postfilter.sh: the last line is the folder
    print(tmp)
    sys.exit(0)
cmd = [os.path.join(ROOT, "hc-mvs_amd", "DensifyPointCloud"), "--input-file", path, "-w", tmp, "-o", os.path.join(tmp, "scene_dense.mvs"), "--verbosity", "2",
       "--fusion-mode", "0", "--max-resolution", "6400", "--min-resolution", "100", "--estimate-normals", "2", "--number-views", "10",
       "--filter-point-cloud", "0", "--resolution-level", "0", "--number-views-fuse", "2", "--n-EstimationIters", "3",
       "--n-EstimationIters-external", "4", "--n-opticalflow", "0", "--n-initTriangulate", "1", "--n-photometric_flow", "0.26", "--n-nOptimize", "1",
       "--n-adapthalfwin", "7", "--n-propagatehalfwin", "5", "--n-propagatestep", "4", "--resume", "0"] + extra
t1 = time.time()
r = subprocess.run(cmd, capture_output=True, text=True)
print(r.stdout)
print(r.stderr[-2000:])
print("driver wall %.2f s, exit %d" % (time.time() - t1, r.returncode))
acc = []
mvsio = importlib.import_module("hc-mvs_amd.mvsio")
for i in range(0, N, max(1, N // 8)):
    dm = mvsio.read_dmap(os.path.join(tmp, "depth%04d.dmap" % i))
    m = dm["depth"] > 0
    gt = views[i]["depth"]
    acc.append((m.mean(), (np.abs(dm["depth"] - gt)[m] / gt[m] < 0.01).mean()))
print("valid fraction %.3f, within 1 %% of ground truth %.3f" % tuple(np.mean(acc, 0)))
