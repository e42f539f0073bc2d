"""The C++ host-side mirror (include/splat_host.hpp, libsplat_host.so): Camera / Gaussian / loader
against the oracle on CPU, the two pipelines' render_to_buffer against the oracle on the GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import splat_amd
from splat_amd import _lib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "splat_amd", "libsplat_host.so")


@pytest.fixture(scope="module")
def H():
    L = C.CDLL(HOST)
    fp = C.POINTER(C.c_float)
    L.splat_host_camera.argtypes = [C.c_float, C.c_float, fp, C.c_float, C.c_float, C.c_int, C.c_float,
                                    C.POINTER(_lib.CameraC)]
    L.splat_host_load_ply.argtypes = [C.c_char_p, fp, fp, fp, fp, fp, C.c_char_p, C.c_int]
    L.splat_host_load_ply.restype = C.c_longlong
    L.splat_host_cov3d.argtypes = [C.c_ulonglong, fp, fp, fp]
    L.splat_host_render.argtypes = [C.c_int, C.c_char_p, C.c_float, C.c_float, fp, C.POINTER(C.c_uint32), C.c_char_p,
                                    C.c_int]
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.mark.parametrize("pos,yaw,pitch", [((0, 0, 5), 0.0, 0.0), ((-0.57651054, 2.99040512, -0.03924271), 0.0, 0.0),
                                           ((0, 0, 5), 10 * np.pi / 180, 0.0), ((1, 2, 3), 1.2, -0.7)])
def test_cpp_camera_bit_equal_to_oracle(H, pos, yaw, pitch):
    out = _lib.CameraC()
    p = np.asarray(pos, np.float32)
    H.splat_host_camera(600.0, 800.0, _fp(p), yaw, pitch, 1, 0.01, C.byref(out))
    oc = O.camera(600, 800, pos, yaw=yaw, pitch=pitch)
    for f in ("view", "proj", "cam_pos"):
        assert list(getattr(out, f)) == list(getattr(oc, f)), f      # same libm, same order: identical bits
    assert (out.w, out.h, out.htanx, out.htany, out.focal) == (oc.w, oc.h, oc.htanx, oc.htany, oc.focal)


def test_cpp_camera_identity_until_pose_update(H):
    out = _lib.CameraC()
    p = np.asarray((0, 0, 5), np.float32)
    H.splat_host_camera(720.0, 1280.0, _fp(p), 0.0, 0.0, 0, 0.3, C.byref(out))
    assert np.array_equal(np.array(out.view[:]).reshape(4, 4), np.eye(4))
    assert np.array_equal(np.array(out.proj[:]).reshape(4, 4), np.eye(4))
    assert out.focal == 360.0 and out.lowpass == np.float32(0.3)


def test_cpp_loader_bit_equal_to_oracle(H, tmp_path):
    raw = splat_amd.gaussians.synthetic_raw(1000, 5)
    p = str(tmp_path / "s.ply")
    splat_amd.write_ply(p, raw, 1000)
    n = H.splat_host_load_ply(p.encode(), None, None, None, None, None, None, 0)
    assert n == 1000
    pos4, sc, op, rot, sh = (np.zeros((n, 4), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32),
                             np.zeros((n, 4), np.float32), np.zeros((n, 48), np.float32))
    assert H.splat_host_load_ply(p.encode(), _fp(pos4), _fp(sc), _fp(op), _fp(rot), _fp(sh), None, 0) == n
    o = O.load_ply(p)
    for got, key in ((pos4, "pos4"), (sc, "scales"), (op, "opacity"), (rot, "rot"), (sh, "sh")):
        assert np.array_equal(got, o[key]), key          # incl. exp / sigmoid / sequential-f32 recentring
    # golden fixture as well
    g = os.path.join(ROOT, "tests", "golden", "c1_head.ply")
    assert H.splat_host_load_ply(g.encode(), None, None, None, None, None, None, 0) == 64


def test_cpp_loader_errors(H, tmp_path):
    p = str(tmp_path / "b.ply")
    open(p, "w").write("ply\nformat ascii 1.0\nelement face 1\nproperty float x\nend_header\n1\n")
    err = C.create_string_buffer(256)
    assert H.splat_host_load_ply(p.encode(), None, None, None, None, None, err, 256) == -1
    assert b"Unexpected element" in err.value
    assert H.splat_host_load_ply(b"/nonexistent.ply", None, None, None, None, None, err, 256) == -1


def test_cpp_cov3d_bit_equal_to_oracle(H):
    g = splat_amd.synthetic_scene(500, 2)
    out = np.zeros((500, 9), np.float32)
    H.splat_host_cov3d(500, _fp(g.scales), _fp(g.rotations), _fp(out))
    assert np.array_equal(out, O.compute_cov3d(g.scales, g.rotations))


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline,lowpass", [(1, 0.01), (2, 0.3)])
def test_cpp_pipelines_render_to_buffer(H, tmp_path, pipeline, lowpass):
    """GaussianSplatPipeline01/02::render_to_buffer (C++) == oracle, on a PLY and on naive_gaussians()"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import image_diff
    raw = splat_amd.gaussians.synthetic_raw(5000, 8)
    ply = str(tmp_path / "s.ply")
    splat_amd.write_ply(ply, raw, 5000)
    pos = np.asarray((0, 0, 5), np.float32)
    for path in (ply, None):
        img = np.zeros((150, 200), np.uint32)
        err = C.create_string_buffer(256)
        rc = H.splat_host_render(pipeline, path.encode() if path else None, 150.0, 200.0, _fp(pos),
                                 img.ctypes.data_as(C.POINTER(C.c_uint32)), err, 256)
        assert rc == 0, err.value
        if path:
            o = O.load_ply(path)
        else:
            g = splat_amd.naive_gaussians()
            o = dict(pos4=g.positions, scales=g.scales, rot=g.rotations, opacity=g.opacities, sh=g.sh)
        scene = dict(pos4=o["pos4"], cov3d=O.compute_cov3d(o["scales"], o["rot"]), opacity=o["opacity"], sh=o["sh"])
        ref, _ = O.render(scene, O.camera(150, 200, (0, 0, 5), lowpass=lowpass))
        assert img.any()
        assert image_diff(img, ref)[0] <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [1, 2])
def test_cpp_render_frame_to_buffer_is_clear_plus_render_to_buffer(H, pipeline):
    """GaussianSplatPipeline01/02::render_frame_to_buffer (the pair at src/main.rs:73-74 as one call, selector 11 / 12 of the
    C shim) writes the image render_to_buffer blends onto zeros, whatever the buffer held"""
    pos = np.asarray((0, 0, 5), np.float32)
    err = C.create_string_buffer(256)
    want = np.zeros((150, 200), np.uint32)
    assert H.splat_host_render(pipeline, None, 150.0, 200.0, _fp(pos), want.ctypes.data_as(C.POINTER(C.c_uint32)), err, 256) == 0, err.value
    got = np.full((150, 200), 0xdeadbeef, np.uint32)
    assert H.splat_host_render(10 + pipeline, None, 150.0, 200.0, _fp(pos), got.ctypes.data_as(C.POINTER(C.c_uint32)), err, 256) == 0, err.value
    assert want.any() and np.array_equal(got, want)


@pytest.mark.gpu
def test_cli_one_call_frames_give_the_same_last_frame(tmp_path):
    """splat_cli --frame: the loop with render_frame_to_buffer into a pinned `color` ends on the frame the literal loop ends on"""
    outs = []
    for extra in ([], ["--frame"]):
        out = str(tmp_path / ("g%d.ppm" % len(outs)))
        r = subprocess.run([os.path.join(ROOT, "splat_amd", "splat_cli"), "--frames", "5", "--size", "160", "120", "--out", out] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.count("Rendering took") == 5
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_cli_main_loop(tmp_path):
    out = str(tmp_path / "f.ppm")
    r = subprocess.run([os.path.join(ROOT, "splat_amd", "splat_cli"), "--frames", "3", "--size", "160", "120", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Rendering took") == 3
    assert os.path.getsize(out) == len("P6\n160 120\n255\n") + 160 * 120 * 3


@pytest.mark.gpu
def test_cli_streaming_loop_gives_the_same_last_frame(tmp_path):
    """--stream: the loop with presentation decoupled (two pinned frames in flight) ends on the frame
    the synchronous loop ends on"""
    outs = []
    for extra in ([], ["--stream"], ["--stream", "--in-flight", "4"]):
        out = str(tmp_path / ("f%d.ppm" % len(outs)))
        r = subprocess.run([os.path.join(ROOT, "splat_amd", "splat_cli"), "--frames", "5", "--size", "160", "120", "--out", out] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert "Streamed 5 frames" in r.stdout
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.gpu
def test_cli_fast_mode_frame_is_within_one_count(tmp_path):
    """--fast = PipelineBase::set_mode(SPLAT_MODE_FAST): every colour byte of the last frame within 1 of the default's"""
    outs = []
    for extra in ([], ["--fast"]):
        out = str(tmp_path / ("f%d.ppm" % len(outs)))
        r = subprocess.run([os.path.join(ROOT, "splat_amd", "splat_cli"), "--frames", "2", "--size", "320", "240", "--out", out] + extra,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        outs.append(np.frombuffer(open(out, "rb").read()[len("P6\n320 240\n255\n"):], np.uint8).astype(np.int16))
    assert outs[0].any() and np.abs(outs[0] - outs[1]).max() <= 1


@pytest.mark.gpu
def test_plain_c_client(tmp_path):
    """examples/render_c.c: a C99 program against include/splat_hip.h only"""
    out = str(tmp_path / "c.ppm")
    r = subprocess.run([os.path.join(ROOT, "splat_amd", "render_c"), out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4 visible" in r.stdout and os.path.getsize(out) == len("P6\n320 240\n255\n") + 320 * 240 * 3
    assert "splat_render_frame: same frame" in r.stdout       # the one-call viewer-loop frame, pageable and page-locked image


GOLD_KAT = None


def _kat():
    global GOLD_KAT
    if GOLD_KAT is None:
        import json
        GOLD_KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "notebook_kat.json")))
    return GOLD_KAT


@pytest.mark.parametrize("ci", range(3))
def test_cpp_and_python_cameras_vs_reference_derived_vectors(H, ci):
    """the C++ camera and the oracle's are the same code typed twice, so "bit equal to the
    oracle" proves little.  Here BOTH host cameras (C++ splat::Camera, Python splat_amd.Camera) are held directly
    against the view / projection matrices and htan/focal that the reference's own Python prototype produced
    (tests/golden/notebook_kat.json, generated by make_notebook_kat.py from notes/util.py)."""
    case = _kat()["cases"][ci]
    want_v, want_p = np.array(case["view"]), np.array(case["proj"])
    # C++ (libsplat_host.so)
    out = _lib.CameraC()
    p = np.asarray(case["pos"], np.float32)
    H.splat_host_camera(float(case["h"]), float(case["w"]), _fp(p), 0.0, 0.0, 1, 0.3, C.byref(out))
    V = np.array(out.view[:]).reshape(4, 4).T            # column-major -> [row][col]
    P = np.array(out.proj[:]).reshape(4, 4).T
    np.testing.assert_allclose(V, want_v, atol=2e-7)
    np.testing.assert_allclose(P, want_p, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose([out.htanx, out.htany, out.focal], case["htanfovxy_focal"], rtol=1e-6)
    # Python (splat_amd/camera.py)
    cam = splat_amd.Camera(case["h"], case["w"], case["pos"])
    cam.update_camera_pose()
    np.testing.assert_allclose(cam.get_view_matrix(), want_v, atol=2e-7)
    np.testing.assert_allclose(cam.get_project_matrix(), want_p, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(cam.get_htanfovxy_focal(), case["htanfovxy_focal"], rtol=1e-6)
    c = cam.to_c(0.3)
    assert list(c.view) == list(out.view) and list(c.proj) == list(out.proj)     # and the two mirrors agree bit for bit
