"""Scene containers and loaders: host-side mirror of src/gaussians.rs.

`GaussianList` keeps the reference's SoA layout (src/gaussians.rs:408-416) -- each Gaussian one
contiguous column -- which is exactly what `splat_upload_scene` takes.
"""
import numpy as np

f32 = np.float32
HALF = f32(0.5)


class GaussianList:
    """positions [n,4] (x,y,z,1) - scales [n,3] - opacities [n] - rotations [n,4] in nalgebra coords
    order (i,j,k,w) - sh [n,48] (f_dc then f_rest, un-transposed) - cov3d [n,9] column-major 3x3."""

    def __init__(self, positions, scales, opacities, rotations, sh, cov3d=None):
        n = len(positions)
        self.positions = np.ascontiguousarray(positions, f32).reshape(n, 4)
        self.scales = np.ascontiguousarray(scales, f32).reshape(n, 3)
        self.opacities = np.ascontiguousarray(opacities, f32).reshape(n)
        self.rotations = np.ascontiguousarray(rotations, f32).reshape(n, 4)
        self.sh = np.ascontiguousarray(sh, f32).reshape(n, 48)
        # Gaussian::new leaves cov3d all-zero until compute_cov3d runs (src/gaussians.rs:254)
        self.cov3d = np.zeros((n, 9), f32) if cov3d is None else np.ascontiguousarray(cov3d, f32).reshape(n, 9)
        self.num_gaussians = n

    def __len__(self):
        return self.num_gaussians

    def compute_cov3d(self, renderer):
        """src/gaussians.rs:446-462, on the GPU (kernel K0 via splat_compute_cov3d)."""
        self.cov3d = renderer.compute_cov3d(self.scales, self.rotations)
        return self

    def subset(self, idx):
        return GaussianList(self.positions[idx], self.scales[idx], self.opacities[idx], self.rotations[idx],
                            self.sh[idx], self.cov3d[idx])


def naive_gaussians():
    """The 4-splat test scene, src/gaussians.rs:319-374 (== notes/util_gau.py:25-60)."""
    pos = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1], [0, 0, 1, 1]], f32)
    scales = np.array([[0.03, 0.03, 0.03], [0.2, 0.03, 0.03], [0.03, 0.2, 0.03], [0.03, 0.03, 0.2]], f32)
    rot = np.tile(np.array([0, 0, 0, 1], f32), (4, 1))           # Quaternion::new(w=1, 0, 0, 0)
    col = np.array([[1, 0, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1]], f32)
    sh = np.zeros((4, 48), f32)
    sh[:, :3] = (col - HALF) / f32(0.28209)
    return GaussianList(pos, scales, np.ones(4, f32), rot, sh)


# ---- INRIA 3DGS PLY schema ---------------------------------------------
PLY_PROPS = (["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] +
             ["f_rest_%d" % i for i in range(45)] + ["opacity"] + ["scale_%d" % i for i in range(3)] +
             ["rot_%d" % i for i in range(4)])
_NP_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
             "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
             "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _activate_and_recentre(raw):
    """set_property activations (src/gaussians.rs:258-282) + mean recentring (:394-402).
    raw: dict name -> f32 array (missing names keep Gaussian::new defaults)."""
    n = len(next(iter(raw.values()))) if raw else 0
    get = lambda k, d=0.0: np.asarray(raw[k], f32) if k in raw else np.full(n, d, f32)  # noqa: E731
    pos = np.stack([get("x"), get("y"), get("z"), np.ones(n, f32)], 1)
    scales = np.stack([np.exp(raw[k], dtype=f32) if k in raw else np.zeros(n, f32)
                       for k in ("scale_0", "scale_1", "scale_2")], 1)
    if "opacity" in raw:
        opac = (f32(1.0) / (f32(1.0) + np.exp(-np.asarray(raw["opacity"], f32), dtype=f32))).astype(f32)
    else:
        opac = np.zeros(n, f32)
    # rot_0 -> w = coords[3]; rot_1..3 -> i,j,k = coords[0..2]; identity when absent
    rot = np.stack([get("rot_1"), get("rot_2"), get("rot_3"), get("rot_0", 1.0)], 1)
    sh = np.zeros((n, 48), f32)
    for i in range(3):
        if "f_dc_%d" % i in raw:
            sh[:, i] = raw["f_dc_%d" % i]
    for k, v in raw.items():
        if k.startswith("f_rest_"):
            idx = int(k[7:])
            if idx > 44:
                raise IndexError("f_rest index %d out of range (the reference panics)" % idx)
            sh[:, 3 + idx] = v
    if n:
        # sequential f32 sum (cumsum is sequential), then / n as f32
        avg = (np.cumsum(pos[:, :3], axis=0, dtype=f32)[-1] / f32(n)).astype(f32)
        pos[:, :3] -= avg
    return GaussianList(pos, scales, opac, rot, sh)


def load_from_ply(filename):
    """load_from_ply, src/gaussians.rs:375-405 -- the C++ host mirror's loader (libsplat_host.so:
    mmap + direct decode, libm exp/sigmoid, sequential-f32 recentring; bit-identical to the oracle's
    reader).  Only `float` properties are consumed; a non-"vertex" element raises (the reference panics)."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsplat_host.so")
    if not os.path.exists(path):
        raise RuntimeError("splat_amd: %s is missing -- run __graft_entry__.build()" % path)
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    L.splat_host_load_ply.argtypes = [C.c_char_p, fp, fp, fp, fp, fp, C.c_char_p, C.c_int]
    L.splat_host_load_ply.restype = C.c_longlong
    err = C.create_string_buffer(512)
    n = L.splat_host_load_ply(str(filename).encode(), None, None, None, None, None, err, 512)
    if n < 0:
        raise ValueError(err.value.decode("utf-8", "replace") or "load_from_ply failed")
    pos4, sc, op = np.zeros((n, 4), f32), np.zeros((n, 3), f32), np.zeros(n, f32)
    rot, sh = np.zeros((n, 4), f32), np.zeros((n, 48), f32)
    g = lambda a: a.ctypes.data_as(fp)  # noqa: E731
    if L.splat_host_load_ply(str(filename).encode(), g(pos4), g(sc), g(op), g(rot), g(sh), err, 512) != n:
        raise ValueError(err.value.decode("utf-8", "replace") or "load_from_ply failed")
    return GaussianList(pos4, sc, op, rot, sh)


def write_ply(filename, raw, n):
    """Write the 62-float INRIA layout (binary little endian).  raw: name -> array; missing -> 0."""
    dt = np.dtype([(p, "<f4") for p in PLY_PROPS])
    data = np.zeros(n, dt)
    for k, v in raw.items():
        data[k] = v
    with open(filename, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n)
        for p in PLY_PROPS:
            f.write(("property float %s\n" % p).encode())
        f.write(b"end_header\n")
        data.tofile(f)


def trim_ply(src, dst, count=3, mode="first", seed=0):
    """Fixture tooling, the counterpart of src/bin/00_ply_load.rs (`trim <in> <out>`: copy the
    first three vertices into a new PLY with the same header).  Works on a binary-little-endian or ascii PLY
    with any property list; the header is kept, only the vertex count changes.
      mode="first":  the first `count` vertices (what the reference's tool does, with count = 3)
      mode="random": `count` vertices drawn without replacement from a seeded generator, kept in file order
                     (a subsample that still looks like the scene: for cutting small goldens from real scenes)
    Returns the number of vertices written."""
    if mode not in ("first", "random"):
        raise ValueError("mode must be 'first' or 'random'")
    with open(src, "rb") as f:
        header = []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % src)
            header.append(line)
            if line.strip() == b"end_header":
                break
        fmt, n, stride = None, None, 0
        sizes = {"char": 1, "uchar": 1, "int8": 1, "uint8": 1, "short": 2, "ushort": 2, "int16": 2, "uint16": 2,
                 "int": 4, "uint": 4, "int32": 4, "uint32": 4, "float": 4, "float32": 4, "double": 8, "float64": 8}
        for k, line in enumerate(header):
            t = line.decode("ascii", "replace").split()
            if t[:1] == ["format"]:
                fmt = t[1]
            elif t[:1] == ["element"]:
                if t[1] != "vertex":
                    raise ValueError("Unexpected element!")
                n = int(t[2])
                count = min(count, n)
                header[k] = ("element vertex %d\n" % count).encode()
            elif t[:1] == ["property"]:
                stride += sizes[t[1]]
        if fmt not in ("ascii", "binary_little_endian") or n is None:
            raise ValueError("%s: unsupported PLY (format %r)" % (src, fmt))
        if mode == "first":
            body = b"".join(f.readline() for _ in range(count)) if fmt == "ascii" else f.read(stride * count)
        else:
            pick = np.sort(np.random.default_rng(seed).choice(n, size=count, replace=False))
            if fmt == "ascii":
                want, rows = set(pick.tolist()), []
                for i in range(n):
                    line = f.readline()
                    if i in want:
                        rows.append(line)
                body = b"".join(rows)
            else:
                payload = np.memmap(src, dtype=np.dtype((np.void, stride)), mode="r", offset=f.tell(), shape=(n,))
                body = np.ascontiguousarray(payload[pick]).tobytes()
                del payload
    with open(dst, "wb") as f:
        f.write(b"".join(header))
        f.write(body)
    return count


def synthetic_raw(n, seed):
    """Seeded stand-in for a trained scene, in PLY (pre-activation) units -- SURVEY.md section 8(d)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = {}
    xyz = np.clip(rng.standard_normal((n, 3)) * 1.5, -6.0, 6.0).astype(f32)
    raw["x"], raw["y"], raw["z"] = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy()
    ls = (rng.standard_normal((n, 3)) * 0.8 - 4.0).astype(f32)
    for i in range(3):
        raw["scale_%d" % i] = ls[:, i].copy()
    q = rng.standard_normal((n, 4)).astype(f32)
    for i in range(4):
        raw["rot_%d" % i] = q[:, i].copy()
    raw["opacity"] = (rng.standard_normal(n) * 2.5).astype(f32)
    dc = rng.standard_normal((n, 3)).astype(f32)
    for i in range(3):
        raw["f_dc_%d" % i] = dc[:, i].copy()
    rest = (rng.standard_normal((n, 45)) * 0.15).astype(f32)
    for i in range(45):
        raw["f_rest_%d" % i] = rest[:, i].copy()
    return raw


def synthetic_surface_raw(n, seed):
    """Seeded stand-in that looks like a TRAINED scene rather than a cloud of blobs, in PLY
    (pre-activation) units: flat, anisotropic Gaussians lying ON thin surfaces -- half on a sphere shell of radius 2,
    a quarter on a ground plane, a quarter on a back wall -- their thin axis along the surface normal, tangential
    scales heavy-tailed (one in ten several times larger), opacities skewed to ~0 and ~1.  A tile then sees hundreds of
    splats within a sliver of depth, most of them opaque: unbounded overdraw with early saturation, which is where a
    real 'truck' differs from the isotropic cloud of synthetic_raw."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k, m = n // 2, n // 4
    pos = np.zeros((n, 3), np.float64)
    nrm = np.zeros((n, 3), np.float64)
    d = rng.standard_normal((k, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos[:k] = (2.0 + rng.normal(0.0, 0.002, (k, 1))) * d; nrm[:k] = d                    # sphere shell, millimetres thick
    pos[k:k + m, 0] = rng.uniform(-4, 4, m); pos[k:k + m, 2] = rng.uniform(-4, 4, m)
    pos[k:k + m, 1] = 1.5 + rng.normal(0.0, 0.001, m); nrm[k:k + m] = (0.0, 1.0, 0.0)   # ground plane (up is -y, src/camera.rs:31)
    r = n - k - m
    pos[k + m:, 0] = rng.uniform(-4, 4, r); pos[k + m:, 1] = rng.uniform(-3, 1.5, r)
    pos[k + m:, 2] = -3.0 + rng.normal(0.0, 0.001, r); nrm[k + m:] = (0.0, 0.0, 1.0)    # back wall
    raw = {}
    raw["x"], raw["y"], raw["z"] = (pos[:, a].astype(f32) for a in range(3))
    big = rng.random(n) < 0.1
    for a in (0, 1):                                                                      # tangential axes: heavy tail
        raw["scale_%d" % a] = np.where(big, rng.normal(-2.2, 0.6, n), rng.normal(-3.6, 0.5, n)).astype(f32)
    raw["scale_2"] = rng.normal(-7.0, 0.3, n).astype(f32)                                 # the thin axis
    # rotation: local z -> surface normal (shortest arc), then a random spin about it.  PLY order: rot_0 = w, rot_1..3 = x, y, z
    w = 1.0 + nrm[:, 2]
    q = np.stack([w, -nrm[:, 1], nrm[:, 0], np.zeros(n)], 1)                              # (w, cross(e_z, n))
    flip = w < 1e-6
    q[flip] = (0.0, 1.0, 0.0, 0.0)                                                        # n = -e_z: half a turn about x
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a = rng.uniform(0, 2 * np.pi, n)
    s = np.stack([np.cos(a / 2), np.zeros(n), np.zeros(n), np.sin(a / 2)], 1)             # spin about local z, applied first
    qw = q[:, 0] * s[:, 0] - q[:, 3] * s[:, 3]
    qx = q[:, 1] * s[:, 0] + q[:, 2] * s[:, 3]
    qy = q[:, 2] * s[:, 0] - q[:, 1] * s[:, 3]
    qz = q[:, 0] * s[:, 3] + q[:, 3] * s[:, 0]
    scale = rng.uniform(0.5, 2.0, n)                                                      # stored un-normalised, as trained PLYs are
    for i, c in enumerate((qw, qx, qy, qz)):
        raw["rot_%d" % i] = (c * scale).astype(f32)
    raw["opacity"] = np.where(rng.random(n) < 0.7, rng.normal(4.0, 1.5, n), rng.normal(-2.0, 1.5, n)).astype(f32)
    dc = rng.standard_normal((n, 3)).astype(f32)
    for i in range(3):
        raw["f_dc_%d" % i] = dc[:, i].copy()
    rest = (rng.standard_normal((n, 45)) * 0.15).astype(f32)
    for i in range(45):
        raw["f_rest_%d" % i] = rest[:, i].copy()
    return raw


def synthetic_surface_scene(n, seed):
    """synthetic_surface_raw pushed through the loader's activations + recentring (no file round trip)."""
    return _activate_and_recentre(synthetic_surface_raw(n, seed))


def synthetic_scene(n, seed):
    """synthetic_raw pushed through the loader's activations + recentring (no file round trip)."""
    return _activate_and_recentre(synthetic_raw(n, seed))
