"""File formats at the boundary of the densify path: the `.mvs` scene (MVSI v5,
Interface.h:212-231, 363-619), the raw 'DR' depth map (Interface.h:634-652, writer DepthMap.cpp:2781-2846) and the
binary little-endian PLY point cloud (PointCloud.cpp:105-240).  Pure struct/numpy code, no third-party readers."""
import struct

import numpy as np

MVSI_VERSION = 5


def _w_str(f, s):
    b = s.encode()
    f.write(struct.pack("<Q", len(b))); f.write(b)


def _r_str(f):
    n, = struct.unpack("<Q", f.read(8))
    return f.read(n).decode()


def write_mvs(path, platforms, images, vertices=(), colors=(), normals=()):
    """platforms: [dict(name, cameras=[dict(name, width, height, K(3,3), R(3,3), C(3))], poses=[dict(R, C)])]
    images: [dict(name, platformID, cameraID, poseID, ID, maskName='')]
    vertices: [dict(X(3), views=[(imageID, confidence)])]; colors: (n,3) u8 B,G,R; normals: (n,3) f32"""
    with open(path, "wb") as f:
        f.write(b"MVSI"); f.write(struct.pack("<II", MVSI_VERSION, 0))
        f.write(struct.pack("<Q", len(platforms)))
        for p in platforms:
            _w_str(f, p["name"])
            f.write(struct.pack("<Q", len(p["cameras"])))
            for c in p["cameras"]:
                _w_str(f, c["name"]); _w_str(f, c.get("bandName", ""))
                f.write(struct.pack("<II", c["width"], c["height"]))
                f.write(np.asarray(c["K"], "<f8").tobytes()); f.write(np.asarray(c["R"], "<f8").tobytes())
                f.write(np.asarray(c["C"], "<f8").tobytes())
            f.write(struct.pack("<Q", len(p["poses"])))
            for q in p["poses"]:
                f.write(np.asarray(q["R"], "<f8").tobytes()); f.write(np.asarray(q["C"], "<f8").tobytes())
        f.write(struct.pack("<Q", len(images)))
        for im in images:
            _w_str(f, im["name"]); _w_str(f, im.get("maskName", ""))
            f.write(struct.pack("<IIII", im["platformID"], im["cameraID"], im["poseID"], im["ID"]))
        f.write(struct.pack("<Q", len(vertices)))
        for v in vertices:
            f.write(np.asarray(v["X"], "<f4").tobytes())
            f.write(struct.pack("<Q", len(v["views"])))
            for iid, conf in v["views"]:
                f.write(struct.pack("<If", iid, conf))
        normals = np.asarray(normals, "<f4").reshape(-1, 3)
        f.write(struct.pack("<Q", len(normals))); f.write(normals.tobytes())
        colors = np.asarray(colors, np.uint8).reshape(-1, 3)
        f.write(struct.pack("<Q", len(colors))); f.write(colors.tobytes())
        for _ in range(3):  # lines, linesNormal, linesColor
            f.write(struct.pack("<Q", 0))
        f.write(np.eye(4, dtype="<f8").tobytes())  # transform (version > 1)


def read_mvs(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"MVSI"
        ver, _ = struct.unpack("<II", f.read(8))
        assert ver == MVSI_VERSION, "only MVSI v5 is handled"
        out = dict(version=ver, platforms=[], images=[], vertices=[])
        for _ in range(struct.unpack("<Q", f.read(8))[0]):
            p = dict(name=_r_str(f), cameras=[], poses=[])
            for _ in range(struct.unpack("<Q", f.read(8))[0]):
                c = dict(name=_r_str(f), bandName=_r_str(f))
                c["width"], c["height"] = struct.unpack("<II", f.read(8))
                c["K"] = np.frombuffer(f.read(72), "<f8").reshape(3, 3).copy()
                c["R"] = np.frombuffer(f.read(72), "<f8").reshape(3, 3).copy()
                c["C"] = np.frombuffer(f.read(24), "<f8").copy()
                p["cameras"].append(c)
            for _ in range(struct.unpack("<Q", f.read(8))[0]):
                p["poses"].append(dict(R=np.frombuffer(f.read(72), "<f8").reshape(3, 3).copy(), C=np.frombuffer(f.read(24), "<f8").copy()))
            out["platforms"].append(p)
        for _ in range(struct.unpack("<Q", f.read(8))[0]):
            im = dict(name=_r_str(f), maskName=_r_str(f))
            im["platformID"], im["cameraID"], im["poseID"], im["ID"] = struct.unpack("<IIII", f.read(16))
            out["images"].append(im)
        for _ in range(struct.unpack("<Q", f.read(8))[0]):
            X = np.frombuffer(f.read(12), "<f4").copy()
            views = [struct.unpack("<If", f.read(8)) for _ in range(struct.unpack("<Q", f.read(8))[0])]
            out["vertices"].append(dict(X=X, views=views))
        n, = struct.unpack("<Q", f.read(8)); out["normals"] = np.frombuffer(f.read(12 * n), "<f4").reshape(-1, 3).copy()
        n, = struct.unpack("<Q", f.read(8)); out["colors"] = np.frombuffer(f.read(3 * n), np.uint8).reshape(-1, 3).copy()
        return out


def image_camera(scene, idx):
    """composed K, R, C of image idx (Interface.h:451-459, Platform.cpp:43-53): R = Rcam Rpose, C = Rpose^T Ccam + Cpose"""
    im = scene["images"][idx]
    p = scene["platforms"][im["platformID"]]
    cam, pose = p["cameras"][im["cameraID"]], p["poses"][im["poseID"]]
    return cam["K"], cam["R"] @ pose["R"], pose["R"].T @ cam["C"] + pose["C"], cam["width"], cam["height"]


def write_dmap(path, depth, K, R, C, d_min, d_max, ids, image_name="", normal=None, conf=None, image_size=None):
    """raw 'DR' depth map (Interface.h:634-652): 28-byte header, name, ids (reference first), K, R, C, maps"""
    h, w = depth.shape
    iw, ih = image_size if image_size else (w, h)
    typ = 1 | (2 if normal is not None else 0) | (4 if conf is not None else 0)
    with open(path, "wb") as f:
        f.write(struct.pack("<HBBIIIIff", 0x5244, typ, 0, iw, ih, w, h, d_min, d_max))
        b = image_name.encode(); f.write(struct.pack("<H", len(b))); f.write(b)
        f.write(struct.pack("<I", len(ids))); f.write(np.asarray(ids, "<u4").tobytes())
        f.write(np.asarray(K, "<f8").tobytes()); f.write(np.asarray(R, "<f8").tobytes()); f.write(np.asarray(C, "<f8").tobytes())
        f.write(np.ascontiguousarray(depth, "<f4").tobytes())
        if normal is not None:
            f.write(np.ascontiguousarray(normal, "<f4").tobytes())
        if conf is not None:
            f.write(np.ascontiguousarray(conf, "<f4").tobytes())


def read_dmap(path):
    with open(path, "rb") as f:
        name, typ, _, iw, ih, w, h, dmin, dmax = struct.unpack("<HBBIIIIff", f.read(28))
        assert name == 0x5244, "not a raw 'DR' depth map"
        n, = struct.unpack("<H", f.read(2)); img = f.read(n).decode()
        n, = struct.unpack("<I", f.read(4)); ids = np.frombuffer(f.read(4 * n), "<u4").copy()
        K = np.frombuffer(f.read(72), "<f8").reshape(3, 3).copy(); R = np.frombuffer(f.read(72), "<f8").reshape(3, 3).copy()
        C = np.frombuffer(f.read(24), "<f8").copy()
        out = dict(image_name=img, ids=ids, K=K, R=R, C=C, d_min=dmin, d_max=dmax, image_size=(iw, ih))
        out["depth"] = np.frombuffer(f.read(4 * w * h), "<f4").reshape(h, w).copy()
        if typ & 2:
            out["normal"] = np.frombuffer(f.read(12 * w * h), "<f4").reshape(h, w, 3).copy()
        if typ & 4:
            out["conf"] = np.frombuffer(f.read(4 * w * h), "<f4").reshape(h, w).copy()
        return out


def write_ply(path, xyz, bgr=None, normal=None):
    """binary little-endian PLY: x y z f32 [nx ny nz f32] [red green blue u8] (PointCloud.cpp:189-240)"""
    n = len(xyz)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normal is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if bgr is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.zeros(n, np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if normal is not None:
        rec["nx"], rec["ny"], rec["nz"] = normal[:, 0], normal[:, 1], normal[:, 2]
    if bgr is not None:
        rec["red"], rec["green"], rec["blue"] = bgr[:, 2], bgr[:, 1], bgr[:, 0]
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n)
        for nm, t in fields:
            f.write(("property %s %s\n" % ("float" if t == "<f4" else "uchar", nm)).encode())
        f.write(b"end_header\n")
        f.write(rec.tobytes())


def read_ply(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        props, n = [], 0
        while True:
            ln = f.readline().strip().decode()
            if ln.startswith("element vertex"):
                n = int(ln.split()[-1])
            elif ln.startswith("property"):
                _, t, nm = ln.split()
                props.append((nm, "<f4" if t in ("float", "float32") else "u1"))
            elif ln == "end_header":
                break
        return np.frombuffer(f.read(), np.dtype(props), count=n)


def write_pgm(path, gray_u8):
    h, w = gray_u8.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h)); f.write(np.ascontiguousarray(gray_u8, np.uint8).tobytes())


def write_ppm(path, rgb_u8):
    h, w, _ = rgb_u8.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h)); f.write(np.ascontiguousarray(rgb_u8, np.uint8).tobytes())
