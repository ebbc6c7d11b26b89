"""Synthetic pinhole scenes for tests and bench.py (seeded, analytic ground truth).

A tilted textured plane with a textured sphere in front of it, seen by a reference camera at the origin
and source cameras on an arc around it.  The texture is a band-limited sum of sinusoids evaluated at the
3-D surface point, so every view sees the same surface pattern.  Conventions follow the reference
(Camera.h:150-151, 299-367): x_cam = R (X - C), pixel = K x_cam / z, pixel centres at integers.
"""
import numpy as np


def look_at(C, target, up=(0.0, -1.0, 0.0)):
    """Rotation R (world->camera) of a camera at C looking at target; camera +z forward, +y down."""
    C = np.asarray(C, np.float64)
    z = np.asarray(target, np.float64) - C
    z /= np.linalg.norm(z)
    x = np.cross(-np.asarray(up, np.float64), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z])


class Scene:
    def __init__(self, seed=2, depth0=10.0, slope=(0.12, -0.08), sphere=(0.6, -0.4, 8.0, 1.4), n_waves=24,
                 min_wavelength=0.03, max_wavelength=1.2, contrast=0.17, amp_exp=0.25):
        rng = np.random.RandomState(seed)
        self.depth0 = depth0
        self.slope = slope
        # plane: z = depth0 + sx*x + sy*y  <=>  n.X = d with n = (-sx,-sy,1)/|.|
        n = np.array([-slope[0], -slope[1], 1.0])
        self.plane_n = n / np.linalg.norm(n)
        self.plane_d = depth0 / np.linalg.norm(n)
        self.sphere_c = np.array(sphere[:3], np.float64)
        self.sphere_r = float(sphere[3])
        lam = np.exp(rng.uniform(np.log(min_wavelength), np.log(max_wavelength), n_waves))
        dirs = rng.normal(size=(n_waves, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        self.freq = (2 * np.pi / lam)[:, None] * dirs
        self.phase = rng.uniform(0, 2 * np.pi, n_waves)
        amp = lam ** amp_exp
        self.amp = contrast * amp / np.sqrt((amp ** 2).sum() / 2)

    def texture(self, P):
        v = np.full(P.shape[:-1], 0.5, np.float64)
        for f, ph, a in zip(self.freq, self.phase, self.amp):
            v += a * np.sin(P @ f + ph)
        return np.clip(v, 0.04, 0.96)

    def render(self, K, R, C, w, h, quantise=True):
        """returns gray (h,w) f32, depth (h,w) f32 camera z, normal (h,w,3) f32 camera space"""
        K = np.asarray(K, np.float64); R = np.asarray(R, np.float64); C = np.asarray(C, np.float64)
        xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        X0 = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)
        d = X0 @ R  # world direction of the z=1 ray: R^T X0
        # plane
        denom = d @ self.plane_n
        tp = (self.plane_d - C @ self.plane_n) / np.where(np.abs(denom) < 1e-12, 1e-12, denom)
        tp = np.where(tp > 0, tp, np.inf)
        # sphere
        oc = C - self.sphere_c
        a = (d * d).sum(-1); b = 2 * (d @ oc); c = oc @ oc - self.sphere_r ** 2
        disc = b * b - 4 * a * c
        ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        ts = np.where(ts > 0, ts, np.inf)
        hit_s = ts < tp
        t = np.where(hit_s, ts, tp)
        P = C + d * t[..., None]
        gray = self.texture(P)
        if quantise:
            gray = np.round(gray * 255.0) / 255.0
        nw = np.where(hit_s[..., None], (P - self.sphere_c) / self.sphere_r, -self.plane_n)
        nc = nw @ R.T
        flip = (nc * X0).sum(-1) > 0
        nc = np.where(flip[..., None], -nc, nc)
        return gray.astype(np.float32), t.astype(np.float32), nc.astype(np.float32)


def make_views(w, h, focal, n_src, seed=2, baseline=(0.05, 0.15), scene=None):
    """Reference camera at the origin + n_src source cameras on an arc.  Returns dict with per-view
    K, R, C (float64), gray (f32), depth/normal ground truth; index 0 is the reference view."""
    # band-limit the texture to the pixel footprint at the nominal depth: wavelengths of 3.5 .. 150 px
    px = 10.0 / focal
    scene = scene or Scene(seed, min_wavelength=3.5 * px, max_wavelength=150 * px)
    rng = np.random.RandomState(seed + 1000)
    K = np.array([[focal, 0, (w - 1) / 2.0], [0, focal, (h - 1) / 2.0], [0, 0, 1]], np.float64)
    target = np.array([0.0, 0.0, scene.depth0])
    views = []
    for i in range(n_src + 1):
        if i == 0:
            C = np.zeros(3)
        else:
            ang = 2 * np.pi * (i - 1) / max(n_src, 1) + 0.3
            bl = scene.depth0 * (baseline[0] + (baseline[1] - baseline[0]) * rng.uniform())
            C = np.array([bl * np.cos(ang), bl * np.sin(ang) * 0.7, 0.05 * bl * rng.uniform(-1, 1)])
        R = look_at(C, target) if i else np.eye(3)
        gray, depth, normal = scene.render(K, R, C, w, h)
        views.append(dict(K=K.copy(), R=R, C=C, gray=gray, depth=depth, normal=normal, width=w, height=h))
    return views


def sparse_points(views, n, seed=5):
    """n world points sampled from the reference view's ground-truth depth (exact)."""
    v = views[0]
    rng = np.random.RandomState(seed)
    h, w = v['depth'].shape
    mx, my = min(8, (w - 1) // 2), min(8, (h - 1) // 2)  # margin; small test images still get points
    xs = rng.randint(mx, max(mx + 1, w - mx), n); ys = rng.randint(my, max(my + 1, h - my), n)
    z = v['depth'][ys, xs].astype(np.float64)
    K = v['K']
    Xc = np.stack([(xs - K[0, 2]) * z / K[0, 0], (ys - K[1, 2]) * z / K[1, 1], z], -1)
    return (Xc @ v['R'] + v['C']).astype(np.float32)
