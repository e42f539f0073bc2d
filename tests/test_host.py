"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol the header
declares, the Camera mirror equals the oracle's restatement, the PLY loader equals the oracle's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import splat_amd
from splat_amd import _lib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "splat_hip.h")).read()
    declared = set(re.findall(r"\b(splat_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 14
    L = _lib.lib()                     # CDLL load works without a GPU
    for name in declared:
        assert hasattr(L, name), name
    assert declared == {s[0] for s in _lib.SYMBOLS}


def test_null_handles_are_refused_without_touching_the_device():
    """entry points called with a NULL context / multi handle / pointer return SPLAT_ERR_INVALID (-1 ... never a crash, never a
    HIP call): runs without a GPU"""
    L = _lib.lib()
    inv = _lib.ERR_INVALID
    assert L.splat_set_frame_overlap(None, 2) == inv
    assert L.splat_multi_set_frame_overlap(None, 2) == inv
    assert L.splat_host_register(None, 16) == inv
    assert L.splat_host_unregister(None) == inv
    assert L.splat_sync(None) == inv
    assert L.splat_set_slab(None, 0, 1) == inv
    assert L.splat_frames_dropped(None) == 0
    assert L.splat_stream(None) is None
    import ctypes as C
    v = C.c_double()
    assert L.splat_set_option(None, _lib.OPT_PIPELINE_DEPTH, 2.0) == inv
    assert L.splat_get_option(None, _lib.OPT_PIPELINE_DEPTH, C.byref(v)) == inv


def test_option_constants_follow_the_header():
    """the ctypes binding's OPT_* are include/splat_hip.h's SPLAT_OPT_*, one for one"""
    hdr = open(os.path.join(ROOT, "include", "splat_hip.h")).read()
    opts = dict((n, int(v)) for n, v in re.findall(r"#define SPLAT_(OPT_[A-Z0-9_]+) (\d+)", hdr))
    assert len(opts) >= 15 and sorted(opts.values()) == list(range(1, len(opts) + 1))
    for n, v in opts.items():
        assert getattr(_lib, n) == v, n
    ffi = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    for n, v in opts.items():
        assert re.search(r"pub const SPLAT_%s: i32 = %d;" % (n, v), ffi), n


def test_struct_sizes_match_header():
    # layouts are plain C; sizes computed by hand from include/splat_hip.h
    assert C.sizeof(_lib.CameraC) == 4 * (16 + 16 + 2 + 3 + 3 + 1 + 1)
    assert C.sizeof(_lib.Record) == 64
    assert C.sizeof(_lib.Config) == 40
    assert C.sizeof(_lib.Stats) == 6 * 8 + 6 * 4 + 8 * 8


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(splat_amd.SplatError):
        splat_amd.Renderer()


@pytest.mark.parametrize("pos,yaw,pitch", [((0, 0, 5), 0.0, 0.0), ((0, 0, 3), 0.0, 0.0),
                                           ((-0.57651054, 2.99040512, -0.03924271), 0.0, 0.0),
                                           ((0, 0, 5), 10 * np.pi / 180, 0.0), ((0, 0, 5), -0.5, 0.3),
                                           ((1, 2, 3), 1.2, -0.7)])
def test_camera_matches_oracle(pos, yaw, pitch):
    cam = splat_amd.Camera(600, 800, pos)
    cam.update_yaw_angle(yaw)
    cam.update_pitch_angle(pitch)
    cam.update_camera_pose()
    assert not cam.is_pose_dirty
    oc = O.camera(600, 800, pos, yaw=yaw, pitch=pitch)
    c = cam.to_c(0.01)
    # sin/cos/tan come from different libms (numpy vs glibc): allow an ulp or two
    np.testing.assert_allclose(np.array(c.view[:]), np.array(oc.view[:]), rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.array(c.proj[:]), np.array(oc.proj[:]), rtol=1e-6)
    assert (c.w, c.h) == (oc.w, oc.h)
    np.testing.assert_allclose([c.htanx, c.htany, c.focal], [oc.htanx, oc.htany, oc.focal], rtol=1e-6)
    assert list(c.cam_pos) == list(oc.cam_pos)          # Q8: the position FIELD, not the orbit eye


def test_camera_defaults_and_identity_before_update():
    cam = splat_amd.Camera(720, 1280)
    assert cam.position.tolist() == [0, 0, 3] and cam.up.tolist() == [0, -1, 0]
    assert np.array_equal(cam.get_view_matrix(), np.eye(4)) and np.array_equal(cam.get_project_matrix(), np.eye(4))
    assert cam.is_pose_dirty
    ht = cam.get_htanfovxy_focal()
    assert ht[1] == 1.0 and ht[2] == 360.0 and abs(ht[0] - 1280 / 720) < 1e-6


def test_naive_gaussians_literals():
    g = splat_amd.naive_gaussians()
    assert len(g) == 4
    assert g.positions[:, :3].tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    assert g.rotations.tolist() == [[0, 0, 0, 1]] * 4
    assert np.allclose(g.sh[0, :3], (np.array([1, 0, 1]) - 0.5) / 0.28209)
    assert not g.cov3d.any()                            # all-zero until compute_cov3d (gaussians.rs:254)


def _write_fixture(tmp_path, n=257, seed=11):
    raw = splat_amd.gaussians.synthetic_raw(n, seed)
    raw["nx"] = np.arange(n, dtype=np.float32)          # ignored properties
    p = str(tmp_path / "scene.ply")
    splat_amd.write_ply(p, raw, n)
    return p, raw


def test_ply_loader_matches_oracle(tmp_path):
    p, raw = _write_fixture(tmp_path)
    g = splat_amd.load_from_ply(p)
    o = O.load_ply(p)
    assert len(g) == 257
    assert np.array_equal(g.positions, o["pos4"])       # incl. sequential-f32 mean recentring
    assert np.array_equal(g.rotations, o["rot"])        # rot_0 -> w (coords[3]), NOT normalised
    assert np.array_equal(g.sh, o["sh"])                # f_rest_k -> sh[3+k], no transpose (Q6)
    assert np.array_equal(g.scales, o["scales"]) and np.array_equal(g.opacities, o["opacity"])   # libm both sides
    assert np.array_equal(g.sh[:, 3], raw["f_rest_0"]) and np.array_equal(g.sh[:, 47], raw["f_rest_44"])
    assert np.array_equal(g.rotations[:, 3], raw["rot_0"]) and np.array_equal(g.rotations[:, 0], raw["rot_1"])


def test_ply_ascii_and_ignored_types(tmp_path):
    p = str(tmp_path / "a.ply")
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                "property double opacity\nproperty float scale_0\nproperty uchar red\nend_header\n"
                "1 2 3 0.5 0.0 7\n3 2 1 0.25 1.0 9\n")
    g = splat_amd.load_from_ply(p)
    o = O.load_ply(p)
    assert np.array_equal(g.positions, o["pos4"]) and g.positions[:, :3].tolist() == [[-1, 0, 1], [1, 0, -1]]
    assert not g.opacities.any() and not o["opacity"].any()     # `double` is not Property::Float: ignored
    np.testing.assert_allclose(g.scales[:, 0], [1.0, np.e], rtol=1e-6)
    assert g.rotations.tolist() == [[0, 0, 0, 1]] * 2           # Quaternion::identity()


def test_ply_unexpected_element_raises(tmp_path):
    p = str(tmp_path / "b.ply")
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement face 1\nproperty float x\nend_header\n1\n")
    with pytest.raises(ValueError):
        splat_amd.load_from_ply(p)
    with pytest.raises(RuntimeError):
        O.load_ply(p)


def test_golden_c1_fixture_loader():
    """tests/golden/c1_head.ply: first 64 vertices of the C1 synthetic scene (seed 1)."""
    p = os.path.join(ROOT, "tests", "golden", "c1_head.ply")
    g = splat_amd.load_from_ply(p)
    o = O.load_ply(p)
    assert len(g) == 64 and np.array_equal(g.positions, o["pos4"]) and np.array_equal(g.sh, o["sh"])


def test_div255_identity():
    """The compositor replaces k/255.0f by fma(k, hi, k*lo) with hi+lo = 1/255; exact for every byte."""
    k = np.arange(256, dtype=np.float32)
    want = k / np.float32(255)
    hi = np.float32(float.fromhex("0x1.010102p-8"))
    lo = np.float32(float.fromhex("-0x1.fdfdfep-33"))
    assert hi == np.float32(1.0 / 255.0)
    t = k * lo                                                           # rounded f32 product
    got = (np.float64(k) * np.float64(hi) + np.float64(t)).astype(np.float32)   # fma: exact product, one rounding
    assert np.array_equal(got, want)
    assert np.array_equal((want * np.float32(255)).astype(np.int32), np.arange(256))    # /255*255 round trip


def test_trim_ply_first_three(tmp_path):
    """fixture tooling (src/bin/00_ply_load.rs): the first three splats, same header"""
    p, raw = _write_fixture(tmp_path, n=50, seed=3)
    out = str(tmp_path / "trim.ply")
    assert splat_amd.trim_ply(p, out) == 3
    a = O.load_ply(out)
    assert a["pos4"].shape[0] == 3
    assert np.array_equal(a["sh"], O.load_ply(p)["sh"][:3])          # payload copied verbatim
    assert np.allclose(a["rot"][:, 3], raw["rot_0"][:3])


def test_trim_ply_random_k(tmp_path):
    """the random-k subsampler -- seeded, without replacement, file order kept, payload verbatim;
    binary and ascii"""
    p, raw = _write_fixture(tmp_path, n=400, seed=4)
    full = O.load_ply(p)
    out = str(tmp_path / "rand.ply")
    assert splat_amd.trim_ply(p, out, count=37, mode="random", seed=9) == 37
    a = O.load_ply(out)
    pick = np.sort(np.random.default_rng(9).choice(400, size=37, replace=False))
    assert len(set(pick.tolist())) == 37
    assert np.array_equal(a["sh"], full["sh"][pick])                     # the chosen vertices, in file order
    assert np.array_equal(a["opacity"], full["opacity"][pick])
    out2 = str(tmp_path / "rand2.ply")
    splat_amd.trim_ply(p, out2, count=37, mode="random", seed=9)
    assert open(out, "rb").read() == open(out2, "rb").read()            # deterministic
    splat_amd.trim_ply(p, out2, count=37, mode="random", seed=10)
    assert open(out, "rb").read() != open(out2, "rb").read()
    assert splat_amd.trim_ply(p, out2, count=1000, mode="random") == 400   # count clamps to the file
    # ascii source
    src = str(tmp_path / "ascii.ply")
    with open(src, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 6\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for i in range(6):
            f.write("%d %d %d\n" % (i, 10 * i, 100 * i))
    dst = str(tmp_path / "ascii_r.ply")
    assert splat_amd.trim_ply(src, dst, count=3, mode="random", seed=1) == 3
    rows = open(dst).read().split("end_header\n")[1].strip().split("\n")
    want = np.sort(np.random.default_rng(1).choice(6, size=3, replace=False))
    assert [int(r.split()[0]) for r in rows] == want.tolist()
    with pytest.raises(ValueError):
        splat_amd.trim_ply(src, dst, mode="middle")


def test_threaded_loader_is_independent_of_the_thread_count(tmp_path):
    """the loader on the fast path decodes straight into the SoA buffers on many host threads;
    the result must not depend on how many (same libm call per value, ONE sequential f32 sum for the recentring)
    and must be the oracle reader's, bit for bit"""
    import ctypes as C
    import os
    raw = splat_amd.gaussians.synthetic_raw(20000, 8)
    p = str(tmp_path / "t.ply")
    splat_amd.write_ply(p, raw, 20000)
    g = splat_amd.load_from_ply(p)                   # all host threads
    a = O.load_ply(p)
    assert np.array_equal(g.positions, a["pos4"]) and np.array_equal(g.scales, a["scales"])
    assert np.array_equal(g.opacities, a["opacity"]) and np.array_equal(g.rotations, a["rot"]) and np.array_equal(g.sh, a["sh"])
    L = C.CDLL(os.path.join(os.path.dirname(splat_amd.__file__), "libsplat_host.so"))
    L.splat_host_time_load.restype = C.c_double
    L.splat_host_time_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.c_char_p, C.c_int]
    n = C.c_longlong()
    for threads in (1, 3, 0):
        assert L.splat_host_time_load(p.encode(), threads, C.byref(n), None, 0) > 0 and n.value == 20000


def test_libm_exp_restatement_constants_and_algorithm():
    """SPLAT_MODE_LIBM_EXP: the kernel's table is 2^(i/32) as glibc tabulates it, and the algorithm with exactly those
    constants reproduces the host libm's expf bit for bit (this container's glibc = the GPU box's: same image)."""
    import ctypes
    import re
    import struct
    from decimal import Decimal, getcontext
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "splat_amd", "csrc", "splat_kernels.hip")).read()
    body = src[src.index("EXP2F_TAB[32] = {"):]
    body = body[:body.index("};")]
    tab = [int(h, 16) for h in re.findall(r"0x([0-9a-f]{16})ull", body)]
    assert len(tab) == 32
    getcontext().prec = 60
    ln2 = Decimal(2).ln()
    for i in range(32):
        want = struct.unpack("<Q", struct.pack("<d", float((ln2 * i / 32).exp())))[0] - (i << 47)
        assert tab[i] == want, i
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    N = 32
    inv = float.fromhex("0x1.71547652b82fep+0") * N
    shift = float.fromhex("0x1.8p+52")
    c0 = float.fromhex("0x1.c6af84b912394p-5") / N / N / N
    c1 = float.fromhex("0x1.ebfce50fac4f3p-3") / N / N
    c2 = float.fromhex("0x1.62e42ff0c52d6p-1") / N
    rng = np.random.default_rng(5)
    xs = np.concatenate([-rng.random(60000) * 12.0, -rng.random(20000) * 87.0, -rng.random(2000) * 1e-3, [0.0, -0.0, -87.0]]).astype(np.float32)
    x = xs.astype(np.float64)
    z = inv * x
    kd = z + shift
    ki = kd.view(np.uint64)
    kd = kd - shift
    r = z - kd
    t = (np.array(tab, np.uint64)[(ki & np.uint64(31)).astype(np.int64)] + (ki << np.uint64(47)))
    s = t.view(np.float64)
    zz = c0 * r + c1
    y = ((zz * (r * r) + (c2 * r + 1.0)) * s).astype(np.float32)
    ref = np.array([libm.expf(float(v)) for v in xs], np.float32)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), int((y != ref).sum())


def test_blend_is_monotone_and_never_expands_a_difference_of_states():
    """The lemma behind both of the compositor's early-outs, checked on the oracle's blend() (the restatement of
    src/pipelines.rs:147-167): for a fixed fragment, blend as a map of the 8-bit state is monotone non-decreasing
    (the [lo, hi] bracket of the exact mode) and maps neighbouring states to equal or neighbouring states
    (SPLAT_MODE_FAST: a state within 1 of the exact one stays within 1).  All 256 states, every alpha fragment()
    can return (0, and [1/255, 0.99]) sampled densely at both ends, colours from tame to absurd."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    alphas = np.concatenate([[0.0, 1.0 / 255.0, 0.99], np.nextafter(np.float32(1.0 / 255.0), np.float32(1), dtype=np.float32)[None],
                             rng.uniform(1.0 / 255.0, 0.99, 150), 1.0 / 255.0 + rng.uniform(0, 1e-3, 40), 0.99 - rng.uniform(0, 1e-3, 40)]).astype(np.float32)
    colours = np.concatenate([[0.0, 1.0, 0.5, -0.25, 1.75, 1e30, -1e30, np.inf, -np.inf, np.nan], rng.uniform(-0.2, 1.2, 12)]).astype(np.float32)
    states = np.arange(256, dtype=np.uint32)
    packed = (states << 16) | (states << 8) | states
    for a in alphas:
        for c in colours[rng.permutation(len(colours))[:6]]:
            with np.errstate(invalid="ignore", over="ignore"):
                frag = np.array([a * c, a * np.float32(0.3), a * np.float32(c * 0.5), a], np.float32)   # fragment(): colour * alpha, alpha
            out = np.array([O.blend(int(p), frag) for p in packed], dtype=np.uint32)
            for sh in (16, 8, 0):
                ch = ((out >> sh) & 0xff).astype(np.int32)
                d = np.diff(ch)
                assert d.min() >= 0 and d.max() <= 1, (float(a), float(c), sh, int(d.min()), int(d.max()))
            if a == 0.0:
                assert np.array_equal(out & 0xffffff, packed & 0xffffff)      # a rejected fragment is the identity on RGB


def test_ply_loader_survives_malformed_files(tmp_path):
    """Truncated files, impossible vertex counts, a corrupted header byte, a missing end_header, a payload of random
    bytes: load_from_ply (the C++ mmap loader) raises ValueError or loads, and never reads past the mapping (each case
    in a child process, so a crash would show as its exit code)."""
    import subprocess, sys
    from splat_amd.gaussians import write_ply, synthetic_raw
    base = str(tmp_path / "base.ply")
    write_ply(base, synthetic_raw(50, 3), 50)
    data = open(base, "rb").read()
    hdr_end = data.index(b"end_header\n") + len(b"end_header\n")
    rng = np.random.default_rng(0)
    child = ("import sys\nsys.path.insert(0, %r)\nimport splat_amd\n"
             "try:\n    print('ok', len(splat_amd.load_from_ply(sys.argv[1])))\n"
             "except ValueError as e:\n    print('raised', str(e)[:60])\n") % os.path.join(os.path.dirname(__file__), "..")
    (tmp_path / "child.py").write_text(child)
    outcomes = []
    for k in range(32):
        kind, d = k % 8, bytearray(data)
        if kind == 0: d = d[:int(rng.integers(0, len(d)))]
        elif kind == 1: d = d[:hdr_end + int(rng.integers(0, len(d) - hdr_end))]
        elif kind == 2: d = d.replace(b"element vertex 50", b"element vertex %d" % int(rng.choice([0, 51, 10**6, 2**31, 2**40, -5])))
        elif kind == 3: d = d.replace(b"property float x", b"property double x")
        elif kind == 4: d = d.replace(b"binary_little_endian", bytes(rng.choice([b"binary_big_endian", b"binary_little_endiax"])))
        elif kind == 5: d[int(rng.integers(0, hdr_end))] = int(rng.integers(0, 256))
        elif kind == 6: d = d.replace(b"end_header\n", b"")
        elif kind == 7: d = d[:hdr_end] + bytes(rng.integers(0, 256, len(d) - hdr_end, dtype=np.uint8))
        f = str(tmp_path / ("c%d.ply" % k))
        open(f, "wb").write(bytes(d))
        r = subprocess.run([sys.executable, str(tmp_path / "child.py"), f], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout.strip(), (kind, r.returncode, r.stderr[-300:])
        outcomes.append((kind, r.stdout.split()[0]))
    assert all(o == "raised" for k, o in outcomes if k in (0, 1, 3, 6))          # truncation / size mismatch / no header end
    assert all(o == "ok" for k, o in outcomes if k == 7)                         # any payload of the right size loads


def test_surface_scene_generator_and_ply_round_trip(tmp_path):
    """the trained-like stand-in (bench WORKLOADS['C3s']): seeded, flat along the surface normal, opacities skewed to the
    ends, and identical whether it goes through write_ply -> load_from_ply (the C++ loader) or straight through the
    activations (the bench leg takes the file route)"""
    from splat_amd import gaussians as G
    from oracle import oracle as O
    n = 3000
    raw = G.synthetic_surface_raw(n, 13)
    assert all(np.array_equal(raw[k], G.synthetic_surface_raw(n, 13)[k]) for k in raw)          # seeded
    g = G.synthetic_surface_scene(n, 13)
    path = str(tmp_path / "s.ply")
    G.write_ply(path, raw, n)
    h = splat_amd.load_from_ply(path)
    for a in ("positions", "scales", "opacities", "rotations", "sh"):
        x, y = getattr(g, a), getattr(h, a)
        if a in ("scales", "opacities"):        # numpy's f32 exp and libm's expf may differ in the last place
            assert np.allclose(x, y, rtol=3e-7, atol=0), a
        else:
            assert np.array_equal(x, y), a
    cov = O.compute_cov3d(h.scales, h.rotations).reshape(n, 3, 3).astype(np.float64)
    pos = np.stack([raw["x"], raw["y"], raw["z"]], 1).astype(np.float64)
    for i in range(0, n // 2, 37):                                # sphere shell: the thin axis is the radial direction
        w, v = np.linalg.eigh(cov[i])
        assert w[0] < 2e-2 * w[1] and abs(v[:, 0] @ (pos[i] / np.linalg.norm(pos[i]))) > 0.999      # thin: sigma ratio < 0.15
    assert (h.opacities > 0.9).mean() > 0.4 and (h.opacities < 0.2).mean() > 0.1
