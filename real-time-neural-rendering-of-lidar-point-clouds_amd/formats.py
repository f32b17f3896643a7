"""Data formats on either side of the projector, host-side mirrors
of the reference's readers.  Pure numpy; nothing here touches the GPU.

  N1  calibration text files  (CameraCalibration.cpp:101-209, README.md:89-103)
      trajectory files        (example/render_trajectory/main.cpp:20-65, README.md:92)
  N2  binary PLY + the 0.25 m block grid  (cloudreader.cpp:8-82,122-177)
  N3  pcd.oct grid cache      (Octreegrid.h:53-114)
"""
import numpy as np

from .camera import CameraCalibration

# ----------------------------------------------------------------------------------------
# N1a: calibration


def load_calibration(path):
    """CameraCalibration::loadCalibration (CameraCalibration.cpp:101-209).

    * file name ending in ``cameras.txt``: first non-comment line of COLMAP's
      ``CAMERA_ID MODEL WIDTH HEIGHT fx fy cx cy k...``; MODEL must be OPENCV or
      OPENCV_FISHEYE.  The reference parses fx, fy, cx, cy as *float* (:123-137), so K holds
      float-rounded values.
    * otherwise the 6-line format ``W H / 3x3 K / distortion line / fisheye flag``
      (:160-206), K parsed as double; the distortion line may use commas.
    Distortion parameters are returned on the object (`m_dists`, `m_fishEye`) but, as in
    the reference, never used by the projector."""
    path = str(path)
    if path.endswith("cameras.txt"):
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                tok = line.split()
                model = tok[1]
                if model not in ("OPENCV", "OPENCV_FISHEYE"):
                    raise ValueError("Unsupported camera model: %s" % model)
                w, h = int(tok[2]), int(tok[3])
                fx, fy, cx, cy = (float(np.float32(v)) for v in tok[4:8])
                cal = CameraCalibration.pinhole(fx, fy, cx, cy, w, h)
                nd = 4 if model == "OPENCV_FISHEYE" else 5
                cal.m_dists = [float(np.float32(v)) for v in tok[8:8 + nd]]
                cal.m_fishEye = model == "OPENCV_FISHEYE"
                return cal
        raise ValueError("No valid camera data found in cameras.txt")
    with open(path) as f:
        lines = f.read().split("\n")
    head = " ".join(lines[:4]).split()
    w, h = int(head[0]), int(head[1])
    K = np.array([float(v) for v in head[2:11]], np.float64).reshape(3, 3)
    dists = [float(v) for v in lines[4].replace(",", " ").split()]
    fish = bool(int(lines[5].split()[0])) if len(lines) > 5 and lines[5].split() else False
    if len(dists) != (4 if fish else 5):
        raise ValueError("%s camera expects %d distortion parameters, got %d"
                         % ("Fisheye" if fish else "Pinhole", 4 if fish else 5, len(dists)))
    cal = CameraCalibration(K, w, h)
    cal.m_dists, cal.m_fishEye = dists, fish
    return cal


def write_calibration_txt(path, cal, dists=(0, 0, 0, 0, 0), fisheye=False):
    """The 6-line custom format of README.md:95-102."""
    K = cal.getIntrinsicsMatrix()
    with open(path, "w") as f:
        f.write("%d %d\n" % (cal.getWidth(), cal.getHeight()))
        for r in range(3):
            f.write(" ".join(repr(float(K[r, c])) for c in range(3)) + "\n")
        f.write(" ".join(repr(float(d)) for d in dists) + "\n")
        f.write("%d\n" % int(fisheye))


def write_cameras_txt(path, cal, camera_id=1, dists=(0, 0, 0, 0, 0)):
    """COLMAP cameras.txt, OPENCV model (README.md:94; parser CameraCalibration.cpp:103-158)."""
    K = cal.getIntrinsicsMatrix()
    with open(path, "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        f.write("%d OPENCV %d %d %r %r %r %r %s\n" % (camera_id, cal.getWidth(), cal.getHeight(), float(K[0, 0]),
                                                      float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                                                      " ".join(repr(float(d)) for d in dists)))


# ----------------------------------------------------------------------------------------
# N1b: trajectories


def quat_to_rot(qw, qx, qy, qz):
    """cv::Quatd(qw,qx,qy,qz).normalize().toRotMat3x3() (main.cpp:37-40)."""
    q = np.array([qw, qx, qy, qz], np.float64)
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float64)


def rot_to_quat(R):
    """-> (qw, qx, qy, qz), qw >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q if q[0] >= 0 else -q


def _pose(R, t):
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return M


def read_trajectory_tum(path):
    """What the reference's example really parses (main.cpp:20-65): one pose per line,
    ``timestamp tx ty tz qx qy qz qw``, camera-to-world; '#' lines and empty lines skipped.
    -> list of world->camera 4x4 matrices (the caller-side ``entry.pose.inv()`` of
    main.cpp:96 applied; parity unpinned: cv::Matx44d::inv is LU-based, numpy's too, but the
    libraries differ)."""
    out = []
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            v = [float(t) for t in line.split()[:8]]
            _, tx, ty, tz, qx, qy, qz, qw = v
            out.append(np.linalg.inv(_pose(quat_to_rot(qw, qx, qy, qz), [tx, ty, tz])))
    return out


def read_trajectory_colmap(path):
    """COLMAP images.txt as README.md:92 documents it: ``IMAGE_ID QW QX QY QZ TX TY TZ CAMERA_ID
    NAME``, already world->camera.  COLMAP writes a second line of 2-D points per image; it is
    skipped when present.  -> list of (world->camera 4x4, name) sorted by IMAGE_ID."""
    items = []
    with open(path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    i = 0
    while i < len(lines):
        ln = lines[i]
        i += 1
        if not ln.strip() or ln.lstrip().startswith("#"):
            continue
        tok = ln.split()
        if len(tok) < 10:
            continue
        iid = int(tok[0])
        qw, qx, qy, qz, tx, ty, tz = (float(t) for t in tok[1:8])
        items.append((iid, _pose(quat_to_rot(qw, qx, qy, qz), [tx, ty, tz]), tok[9]))
        if i < len(lines) and not lines[i].lstrip().startswith("#"):
            nxt = lines[i].split()
            if len(nxt) % 3 == 0 and not (len(nxt) >= 10 and not _is_number(nxt[9])):
                i += 1  # the POINTS2D[] line (possibly empty)
    items.sort(key=lambda t: t[0])
    return [(E, name) for _, E, name in items]


def _is_number(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def write_images_txt(path, poses_w2c, camera_id=1):
    """COLMAP images.txt (two lines per image, the second empty) from world->camera matrices."""
    with open(path, "w") as f:
        f.write("# Image list with two lines of data per image:\n"
                "#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
        for k, E in enumerate(poses_w2c):
            q = rot_to_quat(np.asarray(E)[:3, :3])
            t = np.asarray(E)[:3, 3]
            f.write("%d %s %d frame_%d.png\n\n" % (k + 1, " ".join(repr(float(v)) for v in (*q, *t)), camera_id, k + 1))


def write_trajectory_tum(path, poses_w2c):
    """``timestamp tx ty tz qx qy qz qw`` camera-to-world lines (the order main.cpp:32 reads)."""
    with open(path, "w") as f:
        for k, E in enumerate(poses_w2c):
            c2w = np.linalg.inv(np.asarray(E, np.float64))
            q = rot_to_quat(c2w[:3, :3])
            t = c2w[:3, 3]
            f.write(" ".join(repr(float(v)) for v in (k, t[0], t[1], t[2], q[1], q[2], q[3], q[0])) + "\n")


# ----------------------------------------------------------------------------------------
# N2: PLY + block grid

_PLY_TYPES = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "i2", "ushort": "u2", "int16": "i2",
              "uint16": "u2", "int": "i4", "uint": "u4", "int32": "i4", "uint32": "u4", "float": "f4",
              "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply(path):
    """Vertex x, y, z (+ red, green, blue) of an ascii / binary PLY (the subset tinyply serves
    to cloudreader.cpp:122-177).  -> (xyz float32 [n,3], bgr uint8 [n,3]): like the reference
    loader the colours come back in **B, G, R** order (cloudreader.cpp:168); zeros if absent."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        for need in ("x", "y", "z"):
            if need not in names:
                raise ValueError("Missing vertex positions: %s" % need)
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n, ndmin=2)
            cols = {nm: data[:, i] for i, nm in enumerate(names)}
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(nm, end + ty) for nm, ty in props])
            rec = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
            cols = {nm: rec[nm] for nm in names}
    xyz = np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(np.float32)
    if all(c in cols for c in ("red", "green", "blue")):
        bgr = np.stack([cols["blue"], cols["green"], cols["red"]], axis=1).astype(np.uint8)
    else:
        bgr = np.zeros((n, 3), np.uint8)
    return np.ascontiguousarray(xyz), np.ascontiguousarray(bgr)


def write_ply(path, xyz, rgb):
    """binary_little_endian PLY with float x,y,z + uchar red,green,blue (the layout
    cloudreader.cpp:140-170 expects).  `rgb` is in R, G, B order."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    rgb = np.asarray(rgb, np.uint8).reshape(-1, 3)
    rec = np.empty(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["red"], rec["green"], rec["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n"
                 % len(xyz)).encode("ascii"))
        f.write(rec.tobytes())


class Grid:
    """The reference's ``unordered_map<int, OctreeGrid::Block>`` (Octreegrid.h:16-21) as arrays:
    points sorted by block key, `keys[b]`, `offsets[b]:offsets[b+1]`, `bb_min[b]`, `bb_max[b]`.
    Block order is ascending key (the reference iterates an unordered_map: implementation-
    defined order; the projector's output does not depend on it)."""

    def __init__(self, xyz, colors, keys, offsets, bb_min, bb_max, num_blocks):
        self.xyz, self.colors, self.keys, self.offsets = xyz, colors, keys, offsets
        self.bb_min, self.bb_max, self.num_blocks = bb_min, bb_max, tuple(int(v) for v in num_blocks)

    def __len__(self):
        return len(self.keys)

    def num_points(self):  # OctreeGrid::getNumberPoints (Octreegrid.h:150-159)
        return len(self.xyz)

    def vertex_positions(self):  # OctreeGrid::getVertexPositions (Octreegrid.h:162-170): float4, w = 1
        return np.ascontiguousarray(np.concatenate([self.xyz, np.ones((len(self.xyz), 1), np.float32)], axis=1))

    def vertex_colors(self):  # OctreeGrid::getVertexColors (Octreegrid.h:172-180): uchar4, w = 255
        return np.ascontiguousarray(np.concatenate([self.colors, np.full((len(self.colors), 1), 255, np.uint8)], axis=1))


def compute_grid(xyz, colors, block_size=0.25):
    """computeGrid (cloudreader.cpp:8-82) in the reference's fp32 arithmetic: bounding box
    rounded outwards to whole metres, `block_size` cells, key = x + y*nx + z*nx*ny.
    (Quirk kept: bbMax starts at FLT_MIN, the smallest positive float, cloudreader.cpp:13.)"""
    f32 = np.float32
    xyz = np.ascontiguousarray(xyz, f32).reshape(-1, 3)
    colors = np.ascontiguousarray(colors, np.uint8).reshape(-1, 3)
    bb_min = np.floor(np.minimum(xyz.min(axis=0, initial=np.finfo(f32).max), np.finfo(f32).max)).astype(f32)
    bb_max = np.ceil(np.maximum(xyz.max(axis=0, initial=np.finfo(f32).tiny), np.finfo(f32).tiny)).astype(f32)
    ext = (bb_max - bb_min).astype(f32)
    nb = (ext / f32(block_size)).astype(np.int32)  # int(...) truncation (cloudreader.cpp:39-41)
    with np.errstate(divide="ignore", invalid="ignore"):
        cell = np.floor(((xyz - bb_min).astype(f32) / ext).astype(f32) * nb.astype(f32)).astype(np.int64)
    # Out-of-range cells (a point exactly on the rounded-out maximum, or a degenerate axis) only draw a
    # warning in the reference (cloudreader.cpp:54-55); the key is computed from them as they are (:57-58)
    oob = ((cell < 0) | (cell >= nb)).any(axis=1)
    if oob.any():
        import sys
        first = cell[np.argmax(oob)]
        print("out of bounds: %d, %d, %d  (%d points)" % (first[0], first[1], first[2], int(oob.sum())), file=sys.stderr)
    keys = (cell[:, 0] + cell[:, 1] * nb[0] + cell[:, 2] * nb[0] * nb[1]).astype(np.int64)  # encodeKey (Octreegrid.h:48-50)
    order = np.argsort(keys, kind="stable")
    ukeys, first = np.unique(keys[order], return_index=True)
    offsets = np.concatenate([first, [len(keys)]]).astype(np.int64)
    # decodeKey (Octreegrid.h:116-121): C++ int division and remainder truncate towards zero
    nxy = int(nb[0]) * int(nb[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        kz = np.trunc(ukeys / nxy).astype(np.int64) if nxy else np.zeros_like(ukeys)
        rem = ukeys - kz * nxy
        ky = np.trunc(rem / int(nb[0])).astype(np.int64) if nb[0] else np.zeros_like(ukeys)
        kx = np.fmod(rem, int(nb[0])).astype(np.int64) if nb[0] else np.zeros_like(ukeys)
    size = (ext / nb.astype(f32)).astype(f32)  # bbSize_* (cloudreader.cpp:67-76)
    kxyz = np.stack([kx, ky, kz], axis=1).astype(f32)
    blk_min = (bb_min + kxyz * size).astype(f32)
    blk_max = (bb_min + (kxyz + f32(1)) * size).astype(f32)
    return Grid(xyz[order], colors[order], ukeys.astype(np.int32), offsets, blk_min, blk_max, nb)


# ----------------------------------------------------------------------------------------
# N3: pcd.oct


def write_pcd_oct(path, grid):
    """OctreeGrid::writeOctreeBinary (Octreegrid.h:53-80): 4 ints, then per block: key (int),
    n (size_t), n x 3 f32, n x 3 u8, bbMin 3 f32, bbMax 3 f32."""
    with open(path, "wb") as f:
        f.write(np.array(list(grid.num_blocks) + [len(grid)], np.int32).tobytes())
        for b in range(len(grid)):
            lo, hi = int(grid.offsets[b]), int(grid.offsets[b + 1])
            f.write(np.int32(grid.keys[b]).tobytes())
            f.write(np.uint64(hi - lo).tobytes())
            f.write(np.ascontiguousarray(grid.xyz[lo:hi], np.float32).tobytes())
            f.write(np.ascontiguousarray(grid.colors[lo:hi], np.uint8).tobytes())
            f.write(np.ascontiguousarray(grid.bb_min[b], np.float32).tobytes())
            f.write(np.ascontiguousarray(grid.bb_max[b], np.float32).tobytes())


def read_pcd_oct(path):
    """OctreeGrid::readOctreeBinary (Octreegrid.h:83-114) -> Grid (blocks in file order)."""
    buf = np.fromfile(path, dtype=np.uint8)
    nx, ny, nz, nblocks = np.frombuffer(buf, np.int32, 4, 0)
    pos = 16
    xyz, cols, keys, offs, mins, maxs = [], [], [], [0], [], []
    for _ in range(int(nblocks)):
        keys.append(int(np.frombuffer(buf, np.int32, 1, pos)[0]))
        n = int(np.frombuffer(buf, np.uint64, 1, pos + 4)[0])
        pos += 12
        xyz.append(np.frombuffer(buf, np.float32, 3 * n, pos).reshape(n, 3))
        pos += 12 * n
        cols.append(np.frombuffer(buf, np.uint8, 3 * n, pos).reshape(n, 3))
        pos += 3 * n
        mins.append(np.frombuffer(buf, np.float32, 3, pos))
        maxs.append(np.frombuffer(buf, np.float32, 3, pos + 12))
        pos += 24
        offs.append(offs[-1] + n)
    cat = lambda a, d, w: np.ascontiguousarray(np.concatenate(a)) if a else np.zeros((0, w), d)  # noqa: E731
    return Grid(cat(xyz, np.float32, 3), cat(cols, np.uint8, 3), np.array(keys, np.int32), np.array(offs, np.int64),
                np.array(mins, np.float32).reshape(-1, 3), np.array(maxs, np.float32).reshape(-1, 3), (nx, ny, nz))
