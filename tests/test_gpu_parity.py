"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar (BASELINE.json north_star): every RGBA8 channel within 1 LSB; stage outputs
(records, pixel rectangles, tile lists) exact."""
import copy

import numpy as np
import pytest

import splat_amd
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, image_diff, make_camera

pytestmark = pytest.mark.gpu

TOL_LSB = 1          # north_star: "within 1 ULP per RGBA8 channel"


@pytest.fixture(scope="module")
def R():
    r = splat_amd.Renderer()
    yield r
    r.close()


def gpu_scene(R, n, seed):
    g = splat_amd.synthetic_scene(n, seed)
    g.compute_cov3d(R)
    return g


def render_both(R, g, cam, lowpass, conv_kw=None, init=None, sh_dim=15, nthreads=8):
    h, w = int(cam.h), int(cam.w)
    R.upload(g)
    img = np.zeros((h, w), np.uint32) if init is None else init.copy()
    st = R.render(cam.to_c(lowpass, sh_dim), img)
    ref = np.zeros((h, w), np.uint32) if init is None else init.copy()
    conv = O.default_conventions(**(conv_kw or {}))
    ref, ost = O.render(scene_dict(g), oracle_camera(cam, lowpass, sh_dim), conv, ref, nthreads=nthreads)
    return img, st, ref, ost


def test_cov3d_kernel_bit_exact(R):
    g = splat_amd.synthetic_scene(5000, 21)
    got = R.compute_cov3d(g.scales, g.rotations)
    want = O.compute_cov3d(g.scales, g.rotations)
    assert np.array_equal(got, want)


def test_preprocess_records_exact(R):
    """K1 vs Pipeline::vertex restated: same f32 operation order -> identical records."""
    g = gpu_scene(R, 20000, 1)
    cam = make_camera(256, 256)
    R.upload(g)
    img = np.zeros((256, 256), np.uint32)
    st = R.render(cam.to_c(0.01), img)
    rec = R.records()
    want = O.preprocess(scene_dict(g), oracle_camera(cam, 0.01))
    vis = want["visible"] == 1
    assert st.n_visible == vis.sum()
    assert np.array_equal((rec["px0"] <= rec["px1"]), vis)
    for f in ("cx", "cy", "hx", "hy", "conic", "opacity", "rgb"):
        assert np.array_equal(rec[f][vis], want[f][vis]), f
    assert np.array_equal(rec["depth"], want["depth"])
    for f in ("px0", "px1", "py0", "py1"):
        assert np.array_equal(rec[f][vis], want[f][vis]), f


def test_tile_lists_are_stable_depth_sorted(R):
    g = gpu_scene(R, 30000, 2)
    cam = make_camera(240, 320)
    R.upload(g)
    img = np.zeros((240, 320), np.uint32)
    st = R.render(cam.to_c(0.01), img)
    tiles_x, tiles_y = 20, 15
    off, order = R.tile_lists(tiles_x * tiles_y, st.n_pairs)
    want = O.preprocess(scene_dict(g), oracle_camera(cam, 0.01))
    glob = O.sort(g.positions, np.array(cam.to_c(0.01).view[:], np.float32))
    rank = np.empty(len(g), np.int64)
    rank[glob] = np.arange(len(g))
    assert off[-1] == st.n_pairs and off[0] == 0
    total = 0
    for t in range(tiles_x * tiles_y):
        tx, ty = t % tiles_x, t // tiles_x
        lst = order[off[t]:off[t + 1]]
        v = want["visible"] == 1
        m = v & (want["px0"] // 16 <= tx) & (want["px1"] // 16 >= tx) & (want["py0"] // 16 <= ty) & (want["py1"] // 16 >= ty)
        exp = np.nonzero(m)[0]
        exp = exp[np.argsort(rank[exp], kind="stable")]
        assert np.array_equal(lst, exp), "tile %d" % t
        total += len(exp)
    assert total == st.n_pairs


@pytest.mark.parametrize("n,h,w,seed,lowpass", [
    (10000, 256, 256, 1, 0.01),        # BASELINE configs[0] (C1)
    (10000, 256, 256, 1, 0.3),         # Pipeline02 low-pass
    (50000, 200, 333, 4, 0.01),        # ragged: neither dimension a multiple of 16
    (3000, 17, 40, 6, 0.01),           # tiny target, partial tiles only
])
def test_image_parity_synthetic(R, n, h, w, seed, lowpass):
    g = gpu_scene(R, n, seed)
    cam = make_camera(h, w)
    img, st, ref, ost = render_both(R, g, cam, lowpass)
    mx, cnt = image_diff(img, ref)
    assert st.n_visible == ost.n_visible and st.n_pairs == ost.n_tile_pairs
    assert mx <= TOL_LSB, (mx, cnt)
    assert cnt <= 1e-3 * h * w, (mx, cnt)          # and almost every pixel identical


@pytest.mark.parametrize("yaw,pitch,pos", [(0.0, 0.0, (0, 0, 5)), (0.7, 0.0, (0, 0, 5)), (-1.0, 0.4, (0, 0, 4)),
                                            (0.0, 0.0, (0.3, -0.2, 0.5))])
def test_image_parity_camera_poses(R, yaw, pitch, pos):
    """incl. a camera INSIDE the cloud: huge splats, behind-camera culls, z-clip"""
    g = gpu_scene(R, 8000, 9)
    cam = make_camera(180, 240, pos, yaw, pitch)
    img, st, ref, ost = render_both(R, g, cam, 0.01)
    mx, cnt = image_diff(img, ref)
    assert st.n_visible == ost.n_visible and st.n_pairs == ost.n_tile_pairs
    assert mx <= TOL_LSB, (mx, cnt)


def test_naive_scene_both_pipelines(R):
    g = splat_amd.naive_gaussians()
    cam = make_camera(600, 800)                      # src/main.rs:9-13
    p1 = splat_amd.GaussianSplatPipeline01(g, cam, renderer=R)
    img = np.zeros((600, 800), np.uint32)
    p1.render_to_buffer(img)
    # Pipeline01 with cov3d never computed: cov3d = 0 -> cov2d = 0.01 I (main.rs:24-26 is the caller's job)
    ref, _ = O.render(scene_dict(g), oracle_camera(cam, 0.01))
    assert image_diff(img, ref)[0] <= TOL_LSB
    p2 = splat_amd.GaussianSplatPipeline02(g, cam, renderer=R)     # computes cov3d like from_vec
    assert g.cov3d.any()
    img2 = np.zeros((600, 800), np.uint32)
    p2.render_to_buffer(img2)
    ref2, _ = O.render(scene_dict(g), oracle_camera(cam, 0.3))
    assert image_diff(img2, ref2)[0] <= TOL_LSB
    assert img2.any() and not np.array_equal(img, img2)


def test_identity_matrices_when_pose_never_updated(R):
    """Q20: side binaries never call compute_matrices -> identity view/proj; keep that."""
    g = gpu_scene(R, 2000, 3)
    cam = splat_amd.Camera(120, 160, (0, 0, 5))      # no update_camera_pose()
    img, st, ref, ost = render_both(R, g, cam, 0.01)
    assert st.n_visible == ost.n_visible
    assert image_diff(img, ref)[0] <= TOL_LSB


def test_blends_onto_existing_buffer(R):
    g = gpu_scene(R, 4000, 12)
    cam = make_camera(96, 128)
    rng = np.random.default_rng(0)
    init = rng.integers(0, 2**32, (96, 128), dtype=np.uint64).astype(np.uint32)
    img, st, ref, ost = render_both(R, g, cam, 0.01, init=init)
    mx, cnt = image_diff(img, ref)
    assert mx <= TOL_LSB
    # pixels no quad covers keep their old value, alpha byte included
    untouched = (img == init)
    assert untouched.any() and np.array_equal(untouched, ref == init)


@pytest.mark.parametrize("conv", [dict(y_up=0), dict(sample_half=0), dict(zclip=0), dict(zmin=-1.0)])
def test_convention_switches(conv):
    """every euc convention switch is honoured identically by both sides"""
    r = splat_amd.Renderer(**conv)
    try:
        g = gpu_scene(r, 6000, 14)
        cam = make_camera(128, 160, (0.2, 0.1, 1.5))
        img, st, ref, ost = render_both(r, g, cam, 0.01, conv_kw=conv)
        assert st.n_visible == ost.n_visible
        assert image_diff(img, ref)[0] <= TOL_LSB
    finally:
        r.close()


def test_corrected_projection_mode_matches_the_oracle_flag():
    """SPLAT_MODE_CORRECTED_PROJECTION (section 8(f) rank 2, off by default): K1 and the oracle apply the same
    products in the same order, so records stay bit-identical and the frame within 1 LSB; and the mode
    does change the picture (it is not the reference)."""
    g = splat_amd.synthetic_scene(30000, 23)
    cam = make_camera(208, 304, pos=(0.4, -0.3, 4.0), yaw=0.5, pitch=-0.2)
    imgs = {}
    for mode in (splat_amd.MODE_EXACT, splat_amd.MODE_CORRECTED_PROJECTION):
        r = splat_amd.Renderer(mode=mode)
        try:
            if not g.cov3d.any():
                g.compute_cov3d(r)
            conv = dict(corrected_projection=1) if mode else None
            img, st, ref, ost = render_both(r, g, cam, 0.3, conv_kw=conv)
            rec = r.records()
            want = O.preprocess(scene_dict(g), oracle_camera(cam, 0.3), O.default_conventions(**(conv or {})))
            vis = want["visible"] == 1
            for f in ("cx", "cy", "hx", "hy", "conic", "opacity", "rgb"):
                assert np.array_equal(rec[f][vis], want[f][vis]), (mode, f)
            assert st.n_pairs == ost.n_tile_pairs
            assert image_diff(img, ref)[0] <= TOL_LSB
            imgs[mode] = img
        finally:
            r.close()
    assert not np.array_equal(imgs[0], imgs[1])


def test_sh_degrees(R):
    g = gpu_scene(R, 5000, 15)
    cam = make_camera(128, 128)
    outs = []
    for sh_dim in (3, 12, 15, 27, 48):
        img, st, ref, ost = render_both(R, g, cam, 0.01, sh_dim=sh_dim)
        assert image_diff(img, ref)[0] <= TOL_LSB, sh_dim
        outs.append(img)
    assert np.array_equal(outs[2], outs[3])          # 15 and 27 are both degree 2 (Q5)
    assert not np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[3], outs[4])


def test_empty_and_degenerate_scenes(R):
    cam = make_camera(64, 64)
    empty = splat_amd.GaussianList(np.zeros((0, 4)), np.zeros((0, 3)), np.zeros(0), np.zeros((0, 4)), np.zeros((0, 48)))
    R.upload(empty)
    img = np.full((64, 64), 0x12345678, np.uint32)
    st = R.render(cam.to_c(0.01), img)
    assert st.n_visible == 0 and (img == 0x12345678).all()
    # everything behind the camera / NaN / singular (lowpass 0, zero cov3d): nothing drawn, nothing crashes
    g = splat_amd.synthetic_scene(256, 5)
    g.positions[:64, 2] = 50.0
    g.positions[64:128, 0] = np.nan
    R.upload(g)                                       # cov3d all-zero
    img = np.zeros((64, 64), np.uint32)
    st = R.render(cam.to_c(0.0), img)
    ref, ost = O.render(scene_dict(g), oracle_camera(cam, 0.0))
    assert st.n_visible == ost.n_visible == 0 and st.n_singular == ost.n_singular
    assert not img.any()


def test_non_finite_colours_follow_the_reference(R):
    """SH coefficients of +-inf / NaN.  An ACCEPTED fragment saturates (inf -> 255, -inf and
    NaN -> 0 through `as u8`, src/pipelines.rs:159-161); a REJECTED fragment is (0,0,0,0) in the reference
    (src/pipelines.rs:135-143) and must leave RGB alone -- not 0 * inf = NaN -> 0."""
    g = gpu_scene(R, 3000, 27)
    g.sh[0:600:3, 0] = np.inf
    g.sh[1:600:3, 1] = -np.inf
    g.sh[2:600:3, 2] = np.nan
    g.sh[600:800, :3] = np.inf
    cam = make_camera(160, 208)
    rng = np.random.default_rng(2)
    init = rng.integers(0, 2**32, (160, 208), dtype=np.uint64).astype(np.uint32)
    img, st, ref, ost = render_both(R, g, cam, 0.01, init=init)
    mx, cnt = image_diff(img, ref)
    assert st.n_pairs == ost.n_tile_pairs
    assert mx <= TOL_LSB, (mx, cnt)
    assert cnt <= 1e-3 * img.size, (mx, cnt)


def test_deterministic_and_repeatable(R):
    """atomics place pairs in arbitrary bucket order; the sort makes the frame deterministic"""
    g = gpu_scene(R, 40000, 8)
    cam = make_camera(256, 384)
    R.upload(g)
    frames = []
    for _ in range(3):
        img = np.zeros((256, 384), np.uint32)
        R.render(cam.to_c(0.01), img)
        frames.append(img)
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])


def test_long_tile_lists_all_sort_paths():
    """every Gaussian on the same few tiles: lists > 2048 (big LDS sort), 16385..65536 (sorted as runs
    of 16384 by four workgroups, then merged in two levels) and longer (radix through global memory);
    also exercises pair-buffer growth (tiny initial capacity)."""
    runs_seen = set()
    for n, second in ((40000, 20000), (60000, 8000), (93000, 8000), (130000, 10000)):
        r = splat_amd.Renderer(pair_capacity=1000)
        try:
            g = splat_amd.synthetic_scene(n, 17)
            g.positions[:, :3] *= 0.02                    # squeeze the cloud into ~1 tile
            g.positions[:second, 0] += 0.35               # and a second cluster
            g.compute_cov3d(r)
            cam = make_camera(96, 96)
            img, st, ref, ost = render_both(r, g, cam, 0.01)
            assert st.max_tile_len > 16384, st.max_tile_len
            runs_seen.add(min(5, -(-st.max_tile_len // 16384)))
            assert st.n_pairs == ost.n_tile_pairs
            mx, cnt = image_diff(img, ref)
            assert mx <= TOL_LSB, (n, mx, cnt)
        finally:
            r.close()
    assert {2, 3, 4, 5} & runs_seen >= {3, 4, 5} or runs_seen >= {2, 4, 5}, runs_seen      # merge of 2-3 and of 4 runs, and the global path


@pytest.mark.parametrize("buckets", ["1", "0"])
def test_close_up_camera_many_big_rectangles(buckets):
    """camera inside a dense cloud: most Gaussians cover hundreds of tiles, so the blocks' close-ups go
    through the 64x64-tile cell aggregation (K1 one-pass) / the direct path (two-pass); lists, pair
    count and frame must still be the oracle's"""
    import os
    saved = os.environ.get("SPLAT_BUCKETS")
    os.environ["SPLAT_BUCKETS"] = buckets
    try:
        r = splat_amd.Renderer()
        try:
            g = splat_amd.synthetic_scene(6000, 61)
            g.positions[:, :3] *= 0.25
            g.scales[:] *= 6.0                       # fat splats
            g.compute_cov3d(r)
            cam = make_camera(1088, 1600, pos=(0.05, 0.02, 0.6))      # 100 x 68 tiles: two cells across
            img, st, ref, ost = render_both(r, g, cam, 0.3)
            assert st.n_pairs == ost.n_tile_pairs and st.n_visible == ost.n_visible
            assert st.n_pairs > 150 * st.n_visible, (st.n_pairs, st.n_visible)     # hundreds of tiles per Gaussian
            assert image_diff(img, ref)[0] <= TOL_LSB
            n_tiles = 100 * 68
            off, order = r.tile_lists(n_tiles, st.n_pairs)
            assert off[-1] == st.n_pairs
        finally:
            r.close()
    finally:
        os.environ.pop("SPLAT_BUCKETS", None)
        if saved is not None:
            os.environ["SPLAT_BUCKETS"] = saved


def test_binning_paths_agree():
    """One-pass binning (per-tile regions of the key buffer filled by K1) and two-pass binning (count, scan, emit) must
    give the same lists and the same frame; key buffers that do not fit the byte budget fall back to two-pass on
    their own."""
    import os
    g = splat_amd.synthetic_scene(60000, 41)
    g.positions[:, :3] *= 0.6
    cam = make_camera(200, 328)
    saved = {k: os.environ.get(k) for k in ("SPLAT_BUCKETS", "SPLAT_BUCKET_BYTES")}
    out = {}
    try:
        for name, env in (("one", {}), ("two", {"SPLAT_BUCKETS": "0"}), ("nofit", {"SPLAT_BUCKET_BYTES": "100000"})):
            for k in saved:
                os.environ.pop(k, None)
            os.environ.update(env)
            r = splat_amd.Renderer()
            try:
                if not g.cov3d.any():
                    g.compute_cov3d(r)
                r.upload(g)
                img = np.zeros((200, 328), np.uint32)
                st = r.render(cam.to_c(0.01), img)
                n_tiles = ((328 + 15) // 16) * ((200 + 15) // 16)
                off, order = r.tile_lists(n_tiles, st.n_pairs)
                recs = r.records()
                out[name] = (img, off, order, recs, r.binning_mode(), st.n_pairs, st.max_tile_len)
            finally:
                r.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert out["one"][4] > 0 and out["two"][4] == 0 and out["nofit"][4] == 0, [o[4] for o in out.values()]
    for name in ("two", "nofit"):
        assert np.array_equal(out["one"][0], out[name][0]), name
        assert np.array_equal(out["one"][1], out[name][1]) and np.array_equal(out["one"][2], out[name][2]), name
        assert out["one"][3].tobytes() == out[name][3].tobytes(), name
        assert out["one"][5:] == out[name][5:]
    ref, ost = O.render(scene_dict(g), oracle_camera(cam, 0.01), nthreads=8)
    assert out["one"][5] == ost.n_tile_pairs
    assert image_diff(out["one"][0], ref)[0] <= TOL_LSB


def test_lists_longer_than_any_lds_sort_stay_on_one_pass_binning():
    """every tile owns a REGION of the key buffer sized for its own list, so a list of any length -- 16 385..65 536 keys:
    sorted as runs of 16 384 + merge through the second key buffer; beyond: radix passes through memory -- stays on
    one-pass binning (fixed-stride buckets had to grow for the longest list and gave up beyond 65 536 keys), and a
    scene with short lists afterwards gets a small buffer's worth of regions again"""
    import os
    forced_two_pass = os.environ.get("SPLAT_BUCKETS") == "0"
    r = splat_amd.Renderer()
    try:
        for n, lo, hi in ((40000, 16384, 65536), (130000, 65536, 1 << 30)):
            g = splat_amd.synthetic_scene(n, 17)
            g.positions[:, :3] *= 0.02
            g.compute_cov3d(r)
            cam = make_camera(96, 96)
            img, st, ref, ost = render_both(r, g, cam, 0.01)
            assert lo < st.max_tile_len <= hi, st.max_tile_len
            assert forced_two_pass or r.binning_mode() > 0, r.binning_mode()
            assert st.n_pairs == ost.n_tile_pairs
            assert image_diff(img, ref)[0] <= TOL_LSB
            # ... and again, now with regions from the frame before instead of the bootstrap's
            for _ in range(4):
                img2 = np.zeros_like(img)
                st2 = r.render(cam.to_c(0.01), img2)
                assert np.array_equal(img2, img) and st2.n_pairs == st.n_pairs
        g2 = gpu_scene(r, 20000, 5)
        img, st, ref, ost = render_both(r, g2, make_camera(96, 96), 0.01)
        assert (r.binning_mode() > 0 or forced_two_pass) and image_diff(img, ref)[0] <= TOL_LSB
        now, peak = r.device_bytes()
        assert 0 < now <= peak
    finally:
        r.close()


def test_block_culling_never_changes_a_frame():
    """K1 skips 256-Gaussian blocks whose upload-time bounds cannot reach the slab / target.  The test is
    conservative by construction; check it: culling on == culling off (frame, visible count, pair
    count) for cameras outside, inside and looking away from the cloud, full frame and thin slabs."""
    import os
    g = splat_amd.synthetic_scene(50000, 77)
    g.positions[:25000, :3] *= 0.3                  # a dense core plus a wide halo
    poses = [((0, 0, 5), 0.0, 0.0), ((0, 0, 0.5), 0.3, 0.1), ((2.0, 1.0, 0.2), 2.5, -0.4), ((0, 0, 5), 3.14159, 0.0),
             ((6.0, 0.0, 0.0), 1.2, 0.9)]
    saved = os.environ.get("SPLAT_CULL")
    res = {}
    try:
        for cull in ("1", "0"):
            os.environ["SPLAT_CULL"] = cull
            r = splat_amd.Renderer()
            try:
                if not g.cov3d.any():
                    g.compute_cov3d(r)
                r.upload(g)
                out = []
                for pos, yaw, pitch in poses:
                    cam = make_camera(208, 304, pos=pos, yaw=yaw, pitch=pitch)
                    for slab in ((0, -1), (0, 2), (5, 7), (12, 13)):
                        r.set_slab(*slab)
                        img = np.zeros((208, 304), np.uint32)
                        st = r.render(cam.to_c(0.3), img)
                        out.append((img, st.n_visible, st.n_pairs, st.max_tile_len, st.n_blocks_culled))
                res[cull] = out
            finally:
                r.close()
    finally:
        os.environ.pop("SPLAT_CULL", None)
        if saved is not None:
            os.environ["SPLAT_CULL"] = saved
    assert len(res["1"]) == len(res["0"])
    for k, (a, b) in enumerate(zip(res["1"], res["0"])):
        assert a[1:4] == b[1:4], (k, a[1:], b[1:])
        assert np.array_equal(a[0], b[0]), k
    assert all(o[4] == 0 for o in res["0"]) and sum(o[4] > 0 for o in res["1"]) > len(res["1"]) // 2, [o[4] for o in res["1"]]


def test_streamed_frames_equal_synchronous_frames(R):
    """splat_render_stream (viewer loop: cleared frame, asynchronous copy-out, two frames in flight)
    delivers exactly the frames splat_render produces on a cleared buffer -- pinned and pageable
    destination buffers, an orbit of poses."""
    g = gpu_scene(R, 25000, 29)
    R.upload(g)
    H, W = 176, 240
    poses = [make_camera(H, W, yaw=0.35 * k) for k in range(7)]
    want = []
    for cam in poses:
        img = np.zeros((H, W), np.uint32)
        R.render(cam.to_c(0.01), img)
        want.append(img)
    for pinned in (True, False):
        bufs = [R.host_image(H, W) if pinned else np.empty((H, W), np.uint32) for _ in range(2)]
        for b in bufs:
            b[:] = 0xDEADBEEF                       # the stream path must not blend onto this
        got = []
        for k, cam in enumerate(poses):
            R.render_stream(cam.to_c(0.01), bufs[k & 1])
            if k > 0:
                R.stream_wait(bufs[(k - 1) & 1])
                got.append(bufs[(k - 1) & 1].copy())
        R.stream_wait(bufs[(len(poses) - 1) & 1])
        got.append(bufs[(len(poses) - 1) & 1].copy())
        for k in range(len(poses)):
            assert np.array_equal(got[k], want[k]), (pinned, k)
    with pytest.raises(splat_amd.SplatError):
        R.stream_wait(np.zeros((H, W), np.uint32))      # a buffer no frame was streamed to


def test_streamed_orbit_frames_against_the_oracle(R):
    """VERDICT r2 weak 11 / SURVEY section 8(f)-3: the viewer loop of src/main.rs:53-78 -- 36 poses, 10 degrees of yaw
    per frame, four frames in flight through splat_render_stream -- held DIRECTLY against the oracle (not against
    splat_render of the same library): three of the streamed frames, incl. the first and the last, within 1 LSB."""
    g = gpu_scene(R, 40000, 31)
    R.upload(g)
    H, W = 200, 272
    cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
    cam.update_camera_pose()
    poses, cams = [], []
    for k in range(36):
        poses.append(cam.to_c(0.01))
        cams.append(copy.deepcopy(cam))
        cam.update_yaw_angle(10.0 * np.pi / 180.0)          # Key::Right, src/main.rs:57-60
        cam.update_camera_pose()
    NB = 4
    bufs = [R.host_image(H, W) for _ in range(NB)]
    for b in bufs:
        b[:] = 0xDEADBEEF
    keep = {0: None, 17: None, 35: None}
    d0 = R.frames_dropped()
    for k in range(36):
        if k >= NB:
            R.stream_wait(bufs[k % NB])                      # the frame this buffer held (k - NB) is complete: take it ...
            if (k - NB) in keep:
                keep[k - NB] = bufs[k % NB].copy()
        R.render_stream(poses[k], bufs[k % NB])              # ... before the next one goes into it
    for k in range(36 - NB, 36):
        R.stream_wait(bufs[k % NB])
        if k in keep:
            keep[k] = bufs[k % NB].copy()
    R.sync()
    for k, img in keep.items():
        assert img is not None and img.any() and not (img == 0xDEADBEEF).any()
        ref, ost = O.render(scene_dict(g), oracle_camera(cams[k], 0.01), O.default_conventions(), nthreads=8)
        mx, cnt = image_diff(img, ref)
        assert mx <= TOL_LSB and cnt <= 1e-3 * img.size, (k, mx, cnt)
    assert R.frames_dropped() >= d0


def test_timing_is_sampled_on_asynchronous_frames(R):
    """per-kernel HIP events ride on every n-th asynchronous frame (default 8) and on every frame rendered
    with statistics; splat_get_timing reports how many frames carried them"""
    import torch
    g = gpu_scene(R, 20000, 3)
    R.upload(g)
    cam = make_camera(128, 160).to_c(0.01)
    img = torch.zeros((128, 160), dtype=torch.int32, device="cuda")
    R.render_device(cam, img.data_ptr(), sync=True)
    R.timing(reset=True)
    for _ in range(32):
        R.render_device(cam, img.data_ptr())
    ms, frames = R.timing(reset=True)
    assert 2 <= frames <= 32 and ms["composite"] > 0 and ms["preprocess"] > 0, (frames, ms)
    st = R.render_device(cam, img.data_ptr(), sync=True, want_stats=True)
    ms, frames = R.timing(reset=True)
    assert frames == 1 and st.ms_composite > 0 and abs(ms["composite"] - st.ms_composite) < 1e-3


def test_slabs_equal_full_frame(R):
    """multi-GPU decomposition: tile-row slabs rendered separately == the full frame, byte for byte"""
    g = gpu_scene(R, 30000, 19)
    cam = make_camera(200, 320)                       # 13 tile rows, last one 8 px
    R.upload(g)
    full = np.zeros((200, 320), np.uint32)
    R.render(cam.to_c(0.01), full)
    parts = np.zeros_like(full)
    for (a, b) in ((0, 4), (4, 5), (5, 13)):
        R.set_slab(a, b)
        R.render(cam.to_c(0.01), parts)               # each slab only touches its rows
    R.set_slab(0, -1)
    assert np.array_equal(full, parts)


def test_tile_row_loads_and_balanced_slabs(R):
    """the load estimate used to balance multi-GPU slabs, and byte-equality of a balanced partition"""
    from splat_amd import dist as sdist
    g = gpu_scene(R, 30000, 19)
    cam = make_camera(200, 320)
    R.upload(g)
    cam_c = cam.to_c(0.01)
    R.set_slab(3, 7)                                   # must be ignored by, and survive, the load query
    loads = R.tile_row_loads(cam_c)
    R.set_slab(0, -1)
    full = np.zeros((200, 320), np.uint32)
    st = R.render(cam_c, full)
    assert len(loads) == 13 and int(loads.sum()) == st.n_pairs
    want = O.preprocess(scene_dict(g), oracle_camera(cam, 0.01))
    v = want[want["visible"] == 1]
    for r in range(13):
        m = (v["py0"] // 16 <= r) & (v["py1"] // 16 >= r)
        assert int(loads[r]) == int(((v["px1"][m] // 16) - (v["px0"][m] // 16) + 1).sum())
    for world in (2, 3, 8):
        slabs = sdist.slab_partition_balanced(loads, world)
        parts = np.zeros_like(full)
        for s in slabs:
            R.set_slab(*s)
            R.render(cam_c, parts)
        R.set_slab(0, -1)
        assert np.array_equal(full, parts), world


def test_depth_ties_keep_index_order(R):
    """many Gaussians at exactly the same view depth: the per-tile order must still be the reference's
    stable (index) order -- exercises the radix tie fix-up and its bitonic fallback"""
    g = splat_amd.synthetic_scene(6000, 31)
    g.positions[:, 2] = np.float32(0.25)               # one depth for everyone
    g.positions[:4500, :2] *= 0.02                     # a dense cluster: thousands of ties in one tile
    g.compute_cov3d(R)
    cam = make_camera(128, 128)
    img, st, ref, ost = render_both(R, g, cam, 0.01)
    assert st.n_pairs == ost.n_tile_pairs and st.max_tile_len > 2048
    assert image_diff(img, ref)[0] <= TOL_LSB
    off, order = R.tile_lists(64, st.n_pairs)
    for t in range(64):
        lst = order[off[t]:off[t + 1]].astype(np.int64)
        assert np.all(np.diff(lst) > 0), t             # equal depth everywhere => ascending index


def test_anchor_fixture_on_gpu(R):
    """the committed golden frames (tests/golden/anchor_64.npz) through the HIP path"""
    import os
    a = np.load(os.path.join(os.path.dirname(__file__), "golden", "anchor_64.npz"))
    g = splat_amd.GaussianList(a["positions"], np.zeros((64, 3)), a["opacities"], np.zeros((64, 4)), a["sh"], a["cov3d"])
    R.upload(g)
    from splat_amd import _lib
    for name, lp in (("img_lowpass_0p01", 0.01), ("img_lowpass_0p3", 0.3)):
        c = _lib.CameraC()
        c.view[:] = a["view"].tolist()
        c.proj[:] = a["proj"].tolist()
        c.w = c.h = 64.0
        c.htanx, c.htany, c.focal = 1.0, 1.0, 32.0
        c.cam_pos[:] = a["cam_pos"].tolist()
        c.lowpass, c.sh_dim = lp, 15
        img = np.zeros((64, 64), np.uint32)
        R.render(c, img)
        assert image_diff(img, a[name])[0] <= TOL_LSB, name


def test_early_out_is_exact_for_any_threshold():
    """The compositor's early-out must not change a single byte whatever its transmittance threshold:
    off (0), default, and a reckless 1e-2 that makes thousands of brackets fail and retry."""
    import os
    g = splat_amd.synthetic_scene(120000, 33)
    g.positions[:, :3] *= 0.35                      # dense: hundreds of layers per pixel, long lists
    cam = make_camera(192, 256)
    imgs, fallbacks = [], []
    old = os.environ.get("SPLAT_EARLY_EPS")
    try:
        for eps in ("0", "1e-6", "1e-2", "0.5"):
            os.environ["SPLAT_EARLY_EPS"] = eps
            r = splat_amd.Renderer()
            try:
                if not g.cov3d.any():
                    g.compute_cov3d(r)
                r.upload(g)
                img = np.zeros((192, 256), np.uint32)
                st = r.render(cam.to_c(0.01), img)
                imgs.append(img)
                fallbacks.append(st.n_fallback)
            finally:
                r.close()
    finally:
        if old is None:
            os.environ.pop("SPLAT_EARLY_EPS", None)
        else:
            os.environ["SPLAT_EARLY_EPS"] = old
    assert st.max_tile_len > 1000
    assert fallbacks[0] == 0 and fallbacks[3] > fallbacks[1]        # the reckless thresholds do exercise the retry path
    for k in (1, 2, 3):
        assert np.array_equal(imgs[0], imgs[k]), k
    ref, _ = O.render(scene_dict(g), oracle_camera(cam, 0.01), nthreads=8)
    assert image_diff(imgs[0], ref)[0] <= TOL_LSB


def test_bench_multirank_path_on_shared_gpu():
    """bench.py's N>1 path (balanced slabs, per-rank contexts, gather to rank 0) with two and three
    processes sharing this GPU over gloo: the gathered frame must equal the single-context frame."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n, port in ((2, 29631), (3, 29632)):
        # (the two-rank run also ATTEMPTS the native communicator: RCCL refuses ranks that share a device, and the
        # fallback must say so and carry on -- VERDICT r2 item 4)
        env = dict(os.environ, SPLAT_BENCH_SHARE_GPU="1", SPLAT_BENCH_TRY_NATIVE="1" if n == 2 else "0")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                            "--gpus", str(n), "--steps", "3", "--warmup", "1", "--workload", "C2"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        if n == 2:
            assert "native RCCL gather unavailable" in r.stderr and "ncclCommInitRank" in r.stderr, r.stderr[-1500:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        assert d["n_gpus"] == n and d["scaling"] == "strong"
        assert d["multi_gpu_frame_equals_single_gpu_frame"] is True
        assert d["config"]["n_pairs"] == 946132          # the slabs partition the frame's pairs exactly


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` WITHOUT a launcher around it (how the driver's N=1 command line looks with another N)
    must print the line itself: bench.py re-executes as the torchrun launch of N ranks.  C3 -- BASELINE config 4 at
    workload size -- as 8 ranks sharing this GPU (gloo transport), and the one-process form (splat_multi_*) as 4."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPLAT_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    for extra, n in ((["--gpus", "8"], 8), (["--gpus", "4", "--single-process"], 4)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "C3"] + extra,
                           capture_output=True, text=True, timeout=1200, env=env, cwd=root)
        assert r.returncode == 0, "\n".join([l for l in r.stderr.splitlines() if any(w in l.lower() for w in ("fault", "abort", "error", "assert", "terminate", "what()", "hsa"))][:40]) + r.stderr[-1500:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["metric"] == "frames_per_sec"
        assert d["multi_gpu_frame_equals_single_gpu_frame"] is True
        assert d["config"]["n_pairs"] == 8025623                      # the slabs partition the C3 frame's pairs exactly
        assert len(d["kernel_ms_per_rank"]) == n and all("composite" in k for k in d["kernel_ms_per_rank"])
        assert "roofline_frame" in d and d["roofline_frame"]["bytes_algorithmic"] > 0
        drops = d["config"]["frames_dropped"] if "frames_dropped" in d["config"] else d["frames_dropped"]
        assert drops == 0


def test_near_selection_renders_the_fully_sorted_frame():
    """Near selection (the default): of a tile list of more than 2048 keys only the nearest <= SPLAT_OPT_NEAR_SELECT_KEYS are
    selected by depth and sorted; a wave whose walk needs more reports the tile and the repair launch sorts that list in
    full.  Either way the frame must be the one the full sort gives, byte for byte -- with a generous selection (no tile
    repaired), with one so small that most long tiles are repaired, blending onto a non-zero image (a repaired wave must
    not have touched its pixels), in every mode, both walk flavours, and asynchronously."""
    from splat_amd import _lib as L
    g = splat_amd.synthetic_scene(120000, 77)
    g.positions[:, :3] *= 0.22                                 # dense: thousands of keys per tile
    g.opacities[::3] *= 0.05                                   # ... a third of them faint: deep walks
    cams = [make_camera(184, 264, (0.0, 0.1, 3.0), yaw=0.3), make_camera(184, 264, (0.2, 0.1, 0.5), yaw=1.0, pitch=-0.2)]
    rng = np.random.default_rng(5)
    init = rng.integers(0, 2**32, (184, 264), dtype=np.uint64).astype(np.uint32)
    for mode in (splat_amd.MODE_EXACT, splat_amd.MODE_LIBM_EXP, splat_amd.MODE_FAST):
        r = splat_amd.Renderer(mode=mode)
        try:
            if not g.cov3d.any():
                g.compute_cov3d(r)
            r.upload(g)
            for ci, cam in enumerate(cams):
                cam_c = cam.to_c(0.01)
                frames = {}
                for pair in (0, 1):
                    r.set_option(L.OPT_PAIR_WALK, pair)
                    for near in (0, 2048, 512, 64):
                        r.set_option(L.OPT_NEAR_SELECT_KEYS, near)
                        img = init.copy()
                        st = r.render(cam_c, img)
                        clear = np.zeros((184, 264), np.uint32)
                        r.render(cam_c, clear)
                        frames[(pair, near)] = (img, clear, int(st.n_near_tiles), int(st.n_near_fallback), int(st.max_tile_len))
                        if near == 0:
                            assert st.n_near_tiles == 0 and st.n_near_fallback == 0
                base = frames[(0, 0)]
                assert base[4] > 2048 and base[0].any()
                for key, f in frames.items():
                    if mode & splat_amd.MODE_FAST:
                        # (the fast mode's frame depends on where a walk starts -- within 1 of the exact frame whatever the
                        # start; a selection is scanned to its end where a full list is given up on half way)
                        for a, b in ((f[0], base[0]), (f[1], base[1])):
                            d = np.abs(np.stack([((a >> sh) & 255).astype(np.int32) - ((b >> sh) & 255).astype(np.int32) for sh in (24, 16, 8, 0)]))
                            assert d[0].max() == 0 and d[1:].max() <= 1, (mode, ci, key)
                        continue
                    assert np.array_equal(f[0], base[0]), (mode, ci, key, int((f[0] != base[0]).sum()))
                    assert np.array_equal(f[1], base[1]), (mode, ci, key)
                assert frames[(0, 2048)][2] > 0                                   # tiles were served by their selection ...
                assert frames[(0, 64)][3] > 0                                     # ... and a 64-key selection does send tiles to the repair launch
                assert frames[(0, 64)][3] >= frames[(0, 2048)][3]
            # asynchronous frames through the swap of settings (the repair launch rides behind every near frame)
            r.set_option(L.OPT_NEAR_SELECT_KEYS, 128)
            buf = r.host_image(184, 264)
            for _ in range(4):
                r.render_stream(cams[0].to_c(0.01), buf)
                r.stream_wait(buf)
            want = np.zeros((184, 264), np.uint32)
            r.set_option(L.OPT_NEAR_SELECT_KEYS, 0)
            r.render(cams[0].to_c(0.01), want)
            assert np.array_equal(buf, want)
        finally:
            r.close()
    # against the oracle as well (exact mode, default selection)
    r = splat_amd.Renderer()
    try:
        r.upload(g)
        img = np.zeros((184, 264), np.uint32)
        st = r.render(cams[0].to_c(0.01), img)
        ref, ost = O.render(scene_dict(g), oracle_camera(cams[0], 0.01), nthreads=8)
        assert st.n_pairs == ost.n_tile_pairs and image_diff(img, ref)[0] <= TOL_LSB
        # the debug getter still returns every list in full painter's order (the long lists are sorted on demand)
        n_tiles = ((264 + 15) // 16) * ((184 + 15) // 16)
        offs, order = r.tile_lists(n_tiles, int(st.n_pairs))
        depth = r.records()["depth"]
        for t in range(len(offs) - 1):
            d = depth[order[offs[t]:offs[t + 1]]]
            assert (np.diff(d) >= 0).all(), t
    finally:
        r.close()


def test_options_set_from_code_change_the_schedule_not_the_pixels():
    """splat_set_option: what the SPLAT_* environment variables choose, set through the ABI by a host that cannot reach
    its environment.  Every setting must render the frame the defaults render, byte for byte; the
    getter reports what is in force; a value out of range is refused; an option pinned by its environment variable
    stays the operator's."""
    import os
    from splat_amd import _lib as L
    g = splat_amd.synthetic_scene(90000, 52)
    g.positions[:, :3] *= 0.3                                  # dense: lists beyond 2048 keys, early-out in play
    cam = make_camera(200, 296, (0.0, 0.1, 3.0), yaw=0.3)
    cam_c = cam.to_c(0.01)
    r = splat_amd.Renderer()
    try:
        g.compute_cov3d(r)
        r.upload(g)
        base = np.zeros((200, 296), np.uint32)
        st = r.render(cam_c, base)
        assert st.max_tile_len > 2048 and base.any()
        assert r.get_option(L.OPT_PIPELINE_DEPTH) == 6 and r.get_option(L.OPT_FUSED_SORT_MAX) == 2048
        settings = [[(L.OPT_PIPELINE_DEPTH, 1)], [(L.OPT_PIPELINE_DEPTH, 2)], [(L.OPT_PIPELINE_DEPTH, 4), (L.OPT_FUSED_SORT_MAX, 0)],
                    [(L.OPT_PIPELINE_DEPTH, 6), (L.OPT_FUSED_SORT_MAX, 512), (L.OPT_SORT_IN_COMPOSITOR, 1)],
                    [(L.OPT_SORT_IN_COMPOSITOR, 0), (L.OPT_EARLY_OUT_EPS, 0.0)],
                    [(L.OPT_EARLY_OUT_EPS, 1e-4), (L.OPT_EARLY_OUT_MIN_LIST, 128), (L.OPT_EARLY_OUT_SCAN_EIGHTHS, 8)],
                    [(L.OPT_PAIR_WALK, 1), (L.OPT_BLOCK_CULLING, 0), (L.OPT_REGION_SPARE, 1.0)],
                    [(L.OPT_PAIR_WALK, 0), (L.OPT_ONE_PASS_BINNING, 0), (L.OPT_TIMING_EVERY, 1)],
                    [(L.OPT_ONE_PASS_BINNING, 1), (L.OPT_KEY_BUFFER_BYTES, 1 << 20)],            # too small for regions: two-pass
                    [(L.OPT_KEY_BUFFER_BYTES, float(128 << 30)), (L.OPT_PRIORITY_LIST_LEN, 512), (L.OPT_FRAME_OVERLAP, 2)],
                    [(L.OPT_START_HINTS, 0)], [(L.OPT_START_HINTS, 1)], [(L.OPT_START_HINTS, 2), (L.OPT_FRAME_OVERLAP, 1)]]
        for opts in settings:
            for o, v in opts:
                r.set_option(o, v)
                assert r.get_option(o) == pytest.approx(v), (o, v)
            img = np.zeros((200, 296), np.uint32)
            r.render(cam_c, img)
            assert np.array_equal(img, base), (opts, int((img != base).sum()))
            for _ in range(3):                                  # asynchronous frames through the same settings
                buf = r.host_image(200, 296)
                r.render_stream(cam_c, buf)
                r.stream_wait(buf)
                assert np.array_equal(buf, base), opts
        for o, v in ((L.OPT_PIPELINE_DEPTH, 7), (L.OPT_FUSED_SORT_MAX, 4096), (L.OPT_REGION_SPARE, 0.5), (L.OPT_FRAME_OVERLAP, 3), (L.OPT_START_HINTS, 3), (99, 1), (0, 1)):
            with pytest.raises(splat_amd.renderer.SplatError):
                r.set_option(o, v)
    finally:
        r.close()
    saved = os.environ.get("SPLAT_FUSED_SORT")
    os.environ["SPLAT_FUSED_SORT"] = "256"
    try:
        r = splat_amd.Renderer()
        try:
            r.set_option(L.OPT_FUSED_SORT_MAX, 1024)            # accepted, not applied: the environment pins it
            assert r.get_option(L.OPT_FUSED_SORT_MAX) == 256
            r.upload(g)
            img = np.zeros((200, 296), np.uint32)
            r.render(cam_c, img)
            assert np.array_equal(img, base)
        finally:
            r.close()
    finally:
        os.environ.pop("SPLAT_FUSED_SORT", None)
        if saved is not None:
            os.environ["SPLAT_FUSED_SORT"] = saved


def test_set_option_checks_the_value_first_and_leaves_pinned_options_alone(monkeypatch):
    """an invalid value is SPLAT_ERR_INVALID whether or not the operator pinned the option from the environment,
    and is refused before the context's state is touched; a valid one on a pinned option is accepted and ignored.  Pipeline
    depth 1 leaves no compositor lanes behind; the depth raised again does not bring an overlap of 2 back."""
    from splat_amd import _lib as L
    from splat_amd.renderer import SplatError
    monkeypatch.setenv("SPLAT_START_HINTS", "1")
    r = splat_amd.Renderer()
    try:
        assert r.get_option(L.OPT_START_HINTS) == 1
        r.set_option(L.OPT_START_HINTS, 0)                      # pinned: accepted, the operator's value stays
        assert r.get_option(L.OPT_START_HINTS) == 1
        for opt, bad in ((L.OPT_START_HINTS, 7), (L.OPT_EARLY_OUT_EPS, 2.0), (L.OPT_FRAME_OVERLAP, 3), (L.OPT_HOST_ZERO_COPY, 2),
                         (L.OPT_KEYS_PER_GAUSSIAN, 2), (L.OPT_PIPELINE_DEPTH, 0), (99, 1)):
            with pytest.raises(SplatError) as e:
                r.set_option(opt, bad)
            assert e.value.code == L.ERR_INVALID, (opt, bad)
        g = gpu_scene(r, 20000, 3)
        r.upload(g)
        cam_c = make_camera(160, 240).to_c(0.01)
        want = np.zeros((160, 240), np.uint32)
        r.render(cam_c, want)
        r.set_frame_overlap(2)
        r.set_option(L.OPT_PIPELINE_DEPTH, 1)
        assert r.get_option(L.OPT_FRAME_OVERLAP) == 1
        a = np.zeros((160, 240), np.uint32)
        r.render(cam_c, a)
        r.set_option(L.OPT_PIPELINE_DEPTH, 6)
        assert r.get_option(L.OPT_FRAME_OVERLAP) == 1           # (set it again: documented)
        b = np.zeros((160, 240), np.uint32)
        r.render(cam_c, b)
        r.set_option(L.OPT_KEYS_PER_GAUSSIAN, 16)
        c = np.zeros((160, 240), np.uint32)
        r.render(cam_c, c)
        assert r.binning_mode() == max(1 << 22, 16 * 20000)
        assert np.array_equal(a, want) and np.array_equal(b, want) and np.array_equal(c, want)
    finally:
        r.close()


def test_paired_walk_gives_the_same_bytes():
    """The compositor has two flavours of the exact walk -- one record per step, and two records per step with packed
    f32 math (fewer issue slots for latency-bound waves; chosen per frame by the host) -- which must be the same
    function: every component goes through the same IEEE operations in the same order.  Frames are compared byte for
    byte over short lists, long lists (early-out, bracket, retries), odd and even batch sizes, a non-zero image to
    blend onto and non-finite colours; and both against the oracle."""
    import os
    rng = np.random.default_rng(12)
    cases = []
    g1 = splat_amd.synthetic_scene(40000, 51)
    cases.append((g1, make_camera(200, 296), 0.01))
    g2 = splat_amd.synthetic_scene(90000, 52)
    g2.positions[:, :3] *= 0.25                                  # dense: lists of thousands of keys
    g2.sh[::97, 1] = np.inf
    g2.sh[5::89, 2] = np.nan
    cases.append((g2, make_camera(160, 224, (0.0, 0.1, 3.0), yaw=0.4), 0.3))
    frames = {}
    saved = os.environ.get("SPLAT_PAIR_BLEND")
    try:
        for mode in ("0", "1"):
            os.environ["SPLAT_PAIR_BLEND"] = mode
            r = splat_amd.Renderer()
            try:
                for ci, (g, cam, lp) in enumerate(cases):
                    if not g.cov3d.any():
                        g.compute_cov3d(r)
                    r.upload(g)
                    h, w = int(cam.h), int(cam.w)
                    init = np.random.default_rng(ci).integers(0, 2**32, (h, w), dtype=np.uint64).astype(np.uint32)
                    for variant, start in (("clear", np.zeros((h, w), np.uint32)), ("onto", init)):
                        img = start.copy()
                        st = r.render(cam.to_c(lp), img)
                        frames[(mode, ci, variant)] = (img, st.n_pairs, st.max_tile_len)
            finally:
                r.close()
    finally:
        os.environ.pop("SPLAT_PAIR_BLEND", None)
        if saved is not None:
            os.environ["SPLAT_PAIR_BLEND"] = saved
    for ci, (g, cam, lp) in enumerate(cases):
        for variant in ("clear", "onto"):
            a, b = frames[("0", ci, variant)], frames[("1", ci, variant)]
            assert np.array_equal(a[0], b[0]), (ci, variant, int((a[0] != b[0]).sum()))
        assert frames[("0", ci, "clear")][0].any()
    assert frames[("0", 1, "clear")][2] > 2048                  # the dense case does exercise the long-list paths
    ref, ost = O.render(scene_dict(cases[1][0]), oracle_camera(cases[1][1], 0.3), nthreads=8)
    assert image_diff(frames[("1", 1, "clear")][0], ref)[0] <= TOL_LSB


def test_libm_exp_mode_is_the_oracle_bit_for_bit():
    """SPLAT_MODE_LIBM_EXP: with fragment()'s exponential computed the way the host libm computes it, the HIP frame
    must be the oracle's frame EXACTLY -- every other operation of the path is already the reference's f32
    arithmetic in the reference's order.  This is the evidence behind the 1-LSB budget of the default mode: the
    exponential's last place is the only thing it rounds differently."""
    r = splat_amd.Renderer(mode=splat_amd.MODE_LIBM_EXP)
    try:
        cases = [(40000, 61, make_camera(200, 296), 0.01), (40000, 61, make_camera(200, 296, (0.3, 0.1, 2.0), yaw=0.8, pitch=0.2), 0.3),
                 (90000, 62, make_camera(160, 224, (0.0, 0.1, 3.0)), 0.01)]
        for n, seed, cam, lp in cases:
            g = splat_amd.synthetic_scene(n, seed)
            if seed == 62:
                g.positions[:, :3] *= 0.25                      # dense: long lists, early-out and bracket paths
            g.compute_cov3d(r)
            init = np.random.default_rng(seed).integers(0, 2**32, (int(cam.h), int(cam.w)), dtype=np.uint64).astype(np.uint32)
            img, st, ref, ost = render_both(r, g, cam, lp, init=init)
            assert st.n_pairs == ost.n_tile_pairs
            assert np.array_equal(img, ref), (n, seed, image_diff(img, ref))
    finally:
        r.close()


def test_frame_with_fused_clear_equals_clear_then_render(R):
    """splat_render_frame_device (the viewer loop's clear + render_to_buffer, src/main.rs:73-74, the clear fused into the
    compositor) must give the bytes of a memset followed by splat_render_device: whole frame, a slab (rows outside it
    untouched), a camera that leaves most tiles empty, and an empty scene."""
    g = gpu_scene(R, 30000, 44)
    R.upload(g)
    rng = np.random.default_rng(8)
    h, w = 200, 312
    garbage = rng.integers(0, 2**32, (h, w), dtype=np.uint64).astype(np.uint32)
    for cam, slab in ((make_camera(h, w), None), (make_camera(h, w, (4.0, 0.0, 5.0), yaw=0.2), None), (make_camera(h, w), (3, 9))):
        cam_c = cam.to_c(0.01)
        if slab:
            R.set_slab(*slab)
        want = np.zeros((h, w), np.uint32)
        if slab:
            want[:] = garbage
            want[slab[0] * 16:min(slab[1] * 16, h)] = 0
        d = R.device_image(want)
        R.render_device(cam_c, d, sync=True)
        want = R.device_download(d, h, w)
        d2 = R.device_image(garbage)
        R.render_frame_device(cam_c, d2, sync=True)
        got = R.device_download(d2, h, w)
        R.set_slab(0, -1)
        assert np.array_equal(got, want), (slab, int((got != want).sum()))
        assert want.any()
        for p in (d, d2):
            R.device_free(p)
    empty = splat_amd.GaussianList(np.zeros((0, 4)), np.zeros((0, 3)), np.zeros(0), np.zeros((0, 4)), np.zeros((0, 48)))
    R.upload(empty)
    d = R.device_image(garbage)
    R.render_frame_device(make_camera(h, w).to_c(0.01), d, sync=True)
    assert not R.device_download(d, h, w).any()
    R.device_free(d)


def test_render_frame_host_call_is_clear_plus_render_to_buffer(R):
    """splat_render_frame -- `color.clear(0); render_to_buffer(&mut color)` (src/main.rs:73-74) as one synchronous call whose
    image is written, never read -- gives the bytes of zeros + splat_render (the literal pair) for a pageable image (a copy
    behind the frame), a page-locked one from splat_host_alloc and a caller's array after splat_host_register (the
    compositor stores straight into host memory), with the zero copy switched off, on a 36-pose orbit against the oracle,
    for a slab (rows outside it untouched) and for an empty scene."""
    from splat_amd import _lib as L
    g = gpu_scene(R, 30000, 45)
    R.upload(g)
    rng = np.random.default_rng(9)
    h, w = 200, 312
    garbage = rng.integers(0, 2**32, (h, w), dtype=np.uint64).astype(np.uint32)
    cam = make_camera(h, w)
    cam_c = cam.to_c(0.01)
    want = np.zeros((h, w), np.uint32)
    st0 = R.render(cam_c, want)
    assert want.any()
    pinned = R.host_image(h, w)
    reg = garbage.copy()
    splat_amd.Renderer.host_register(reg)
    try:
        for zero_copy in (1, 0):
            R.set_option(L.OPT_HOST_ZERO_COPY, zero_copy)
            assert R.get_option(L.OPT_HOST_ZERO_COPY) == zero_copy
            for name, buf in (("pageable", garbage.copy()), ("host_alloc", pinned), ("registered", reg)):
                buf[:] = garbage
                st = R.render_frame(cam_c, buf, want_stats=True)
                assert np.array_equal(buf, want), (zero_copy, name, int((buf != want).sum()))
                assert (st.n_pairs, st.n_visible) == (st0.n_pairs, st0.n_visible)
                buf[:] = garbage
                assert R.render_frame(cam_c, buf) is None          # the loop's form: no statistics
                assert np.array_equal(buf, want), (zero_copy, name)
        R.set_option(L.OPT_HOST_ZERO_COPY, 1)
        # the reference's loop (src/main.rs:43-78): a 10-degree yaw step, then the frame; every fourth pose against the oracle
        orbit = splat_amd.Camera(h, w, (0.0, 0.0, 5.0))
        for k in range(36):
            orbit.update_camera_pose()
            for buf in (pinned, reg):
                buf[:] = garbage
                R.render_frame(orbit.to_c(0.01), buf)
            assert np.array_equal(pinned, reg), k
            if k % 4 == 0:
                ref, _ = O.render(scene_dict(g), oracle_camera(orbit, 0.01), O.default_conventions(), np.zeros((h, w), np.uint32), nthreads=8)
                assert image_diff(pinned, ref)[0] <= TOL_LSB, (k, image_diff(pinned, ref))
            orbit.update_yaw_angle(10.0 * np.pi / 180.0)
        # a slab: only its rows are written
        R.set_slab(3, 9)
        for buf in (garbage.copy(), pinned):
            buf[:] = garbage
            R.render_frame(cam_c, buf)
            assert np.array_equal(buf[:48], garbage[:48]) and np.array_equal(buf[144:], garbage[144:])
            assert np.array_equal(buf[48:144], want[48:144])
        R.set_slab(0, -1)
        empty = splat_amd.GaussianList(np.zeros((0, 4)), np.zeros((0, 3)), np.zeros(0), np.zeros((0, 4)), np.zeros((0, 48)))
        R.upload(empty)
        for buf in (garbage.copy(), pinned):
            buf[:] = garbage
            R.render_frame(cam_c, buf)
            assert not buf.any()
    finally:
        splat_amd.Renderer.host_unregister(reg)
        R.set_slab(0, -1)


def test_render_frame_into_a_partly_locked_image_takes_the_copy_path(R):
    """the zero-copy frame looked at the FIRST byte of the image only.  An image of which the caller page-locked
    less than W*H*4 bytes (a slab's worth of a larger frame) must be rendered through the copy path -- a store past the
    mapping is a GPU page fault that ends the context -- and an image locked by somebody else's allocator (a torch pinned
    tensor) is asked of the runtime and still served, one way or the other."""
    import ctypes as C
    import torch
    from splat_amd import _lib as L
    g = gpu_scene(R, 20000, 46)
    R.upload(g)
    h, w = 208, 320
    cam_c = make_camera(h, w).to_c(0.01)
    want = np.zeros((h, w), np.uint32)
    R.render(cam_c, want)
    assert want.any()
    R.set_option(L.OPT_HOST_ZERO_COPY, 1)
    # page-aligned storage, the first 64 rows locked (a multiple of the page size: 64 * 320 * 4 = 20 pages)
    raw = np.zeros(h * w + 1024, np.uint32)
    off = (-raw.ctypes.data % 4096) // 4
    img = raw[off:off + h * w].reshape(h, w)
    part = img[:64]
    assert L.lib().splat_host_register(C.c_void_p(part.ctypes.data), part.nbytes) == 0
    try:
        img[:] = 0xdeadbeef
        R.render_frame(cam_c, img)
        assert np.array_equal(img, want)
        # the context is alive and the fully locked form still takes the zero-copy path's result
        img[:] = 0xdeadbeef
        R.render_frame(cam_c, img)
        assert np.array_equal(img, want)
    finally:
        L.lib().splat_host_unregister(C.c_void_p(part.ctypes.data))
    t = torch.empty((h, w), dtype=torch.int32).pin_memory()
    arr = t.numpy().view(np.uint32)
    arr[:] = 0xdeadbeef
    R.render_frame(cam_c, arr)
    assert np.array_equal(arr, want)
    # a view INTO a larger pinned tensor whose tail is not part of it: rows 0..h of a taller image
    big = torch.empty((h + 8, w), dtype=torch.int32).pin_memory()
    sub = big.numpy().view(np.uint32)[8:]
    sub[:] = 0xdeadbeef
    R.render_frame(cam_c, sub)
    assert np.array_equal(sub, want)


def _channels(a):
    return np.stack([(a >> s) & 0xff for s in (24, 16, 8, 0)]).astype(np.int32)


def test_fast_mode_is_within_one_count_of_the_exact_frame():
    """SPLAT_MODE_FAST closes the early-out's bracket at hi - lo <= 2 and continues from its middle; blend() never
    expands a difference of states, so every R, G, B byte must end within 1 of the exact frame's and the alpha byte
    must be the same -- on short lists, dense scenes (thousands of keys per tile), a non-zero
    image to blend onto, both walks (one and two records per step), both bracket widths, and reckless start thresholds
    that make most brackets fail and retry.  With the libm exponential on top the bound holds against the ORACLE."""
    import os
    cases = []
    g1 = splat_amd.synthetic_scene(40000, 71)
    cases.append((g1, make_camera(200, 296), 0.01))
    g2 = splat_amd.synthetic_scene(120000, 72)
    g2.positions[:, :3] *= 0.3                                   # dense: hundreds of layers per pixel
    g2.sh[::101, 1] = np.inf
    cases.append((g2, make_camera(192, 256, (0.0, 0.1, 3.0), yaw=0.3), 0.3))
    keys = ("SPLAT_EARLY_EPS", "SPLAT_FAST_WIDTH", "SPLAT_PAIR_BLEND")
    saved = {k: os.environ.get(k) for k in keys}

    def frames(mode, env):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        r = splat_amd.Renderer(mode=mode)
        out = []
        try:
            for ci, (g, cam, lp) in enumerate(cases):
                if not g.cov3d.any():
                    g.compute_cov3d(r)
                r.upload(g)
                h, w = int(cam.h), int(cam.w)
                init = np.random.default_rng(ci).integers(0, 2**32, (h, w), dtype=np.uint64).astype(np.uint32)
                for start in (np.zeros((h, w), np.uint32), init):
                    img = start.copy()
                    st = r.render(cam.to_c(lp), img)
                    out.append((img, st))
        finally:
            r.close()
        return out

    try:
        exact = frames(splat_amd.MODE_EXACT, {})
        assert exact[2][1].max_tile_len > 2048
        differing = 0
        for env in ({}, {"SPLAT_FAST_WIDTH": "1"}, {"SPLAT_PAIR_BLEND": "1"}, {"SPLAT_PAIR_BLEND": "0"},
                    {"SPLAT_EARLY_EPS": "0.05"}, {"SPLAT_EARLY_EPS": "0.5", "SPLAT_PAIR_BLEND": "1"}, {"SPLAT_EARLY_EPS": "1e-5"}):
            fast = frames(splat_amd.MODE_FAST, env)
            for k, ((a, sa), (b, sb)) in enumerate(zip(exact, fast)):
                d = np.abs(_channels(a) - _channels(b))
                assert d[0].max() == 0, (env, k)                   # the alpha byte: the nearest covering record's, exact
                assert d[1:].max() <= 1, (env, k, int(d[1:].max()))
                assert sa.n_pairs == sb.n_pairs
                differing += int((d.max(0) > 0).sum())
        assert differing > 0                                        # the dense case does take the shortcut
        # ... and against the oracle, with the exponential that makes the exact mode bit-identical to it
        fast = frames(splat_amd.MODE_FAST | splat_amd.MODE_LIBM_EXP, {})
        for ci, (g, cam, lp) in enumerate(cases):
            ref, ost = O.render(scene_dict(g), oracle_camera(cam, lp), nthreads=8)
            d = np.abs(_channels(fast[2 * ci][0]) - _channels(ref))
            assert d.max() <= 1 and d[0].max() == 0, (ci, int(d.max()))
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_compositor_sorts_long_lists_itself():
    """SPLAT_SORT_IN_COMP=1 (chosen automatically when the average list is longer than 2048 keys): lists of more than
    2048 keys are ordered by their tile's compositor workgroup -- depth partition through the second key buffer, parts
    sorted through LDS -- instead of by sort launches.  Same lists, same frame: lists of 2049..16384 keys, longer ones,
    more than 63 parts (the global route), thousands of Gaussians at ONE depth (a bin of more than 512 keys: the global
    route and its tie fix-up), and one-pass as well as two-pass binning."""
    import os
    saved = {k: os.environ.get(k) for k in ("SPLAT_SORT_IN_COMP", "SPLAT_BUCKETS")}
    try:
        for buckets in ("1", "0"):
            os.environ["SPLAT_SORT_IN_COMP"] = "1"
            os.environ["SPLAT_BUCKETS"] = buckets
            r = splat_amd.Renderer()
            try:
                for n, squeeze, second in ((30000, 0.05, 9000), (60000, 0.02, 8000), (130000, 0.02, 10000)):
                    g = splat_amd.synthetic_scene(n, 19)
                    g.positions[:, :3] *= squeeze
                    g.positions[:second, 0] += 0.35
                    g.compute_cov3d(r)
                    cam = make_camera(96, 96)
                    img, st, ref, ost = render_both(r, g, cam, 0.01)
                    assert st.max_tile_len > 2048 and st.n_pairs == ost.n_tile_pairs
                    assert image_diff(img, ref)[0] <= TOL_LSB, (buckets, n)
                g = splat_amd.synthetic_scene(6000, 31)
                g.positions[:, 2] = np.float32(0.25)               # one depth for everyone
                g.positions[:4500, :2] *= 0.02
                g.compute_cov3d(r)
                img, st, ref, ost = render_both(r, g, make_camera(128, 128), 0.01)
                assert st.max_tile_len > 2048 and image_diff(img, ref)[0] <= TOL_LSB
                off, order = r.tile_lists(64, st.n_pairs)
                for t in range(64):
                    assert np.all(np.diff(order[off[t]:off[t + 1]].astype(np.int64)) > 0), t
            finally:
                r.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_fuzzed_scenes_match_the_oracle():
    """tools/fuzz_parity.py's generator (hostile scales, anisotropy, opacities, depth clusters, NaN / inf positions and
    colours, degenerate quaternions, odd target sizes, cameras inside / behind / far, every sorting / binning /
    compositing variant): 60 seeds here, thousands when the tool is run by hand.  HIP == oracle within 1 LSB with equal
    pair and visible counts; the fast mode within 1 of the exact frame with equal alpha bytes; the libm-exp mode the
    oracle's frame bit for bit."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import fuzz_parity as F
    saved = {k: os.environ.get(k) for k in F.KEYS}
    try:
        for seed in range(9000, 9060):
            g, cam, lp, variant, init, desc = F.make_case(seed)
            for k in F.KEYS:
                os.environ.pop(k, None)
            os.environ.update(variant)
            frames = {}
            for mode in (0, splat_amd.MODE_FAST, splat_amd.MODE_LIBM_EXP, splat_amd.MODE_FAST | splat_amd.MODE_LIBM_EXP):
                r = splat_amd.Renderer(mode=mode)
                try:
                    if mode == 0:
                        g.compute_cov3d(r)
                    r.upload(g)
                    img = init.copy()
                    st = r.render(cam.to_c(lp, F.SH_DIMS[seed % len(F.SH_DIMS)]), img)
                    frames[mode] = (img, st)
                finally:
                    r.close()
            sd = scene_dict(g)
            keep = np.isfinite(g.positions).all(axis=1)      # NaN depths make the reference's global sort order undefined
            if not keep.all():
                sd = {k: np.ascontiguousarray(v[keep]) for k, v in sd.items()}
            ref, ost = O.render(sd, oracle_camera(cam, lp, F.SH_DIMS[seed % len(F.SH_DIMS)]), O.default_conventions(), init.copy(), nthreads=8)
            img, st = frames[0]
            assert st.n_pairs == ost.n_tile_pairs and st.n_visible == ost.n_visible, desc
            assert image_diff(img, ref)[0] <= TOL_LSB, desc
            d = np.abs(_channels(frames[splat_amd.MODE_FAST][0]) - _channels(img))
            assert d[0].max() == 0 and d[1:].max() <= 1, desc
            # the exponential as the host libm computes it: the oracle's frame bit for bit; the fast mode within 1 of it
            assert np.array_equal(frames[splat_amd.MODE_LIBM_EXP][0], ref), desc
            assert image_diff(frames[splat_amd.MODE_FAST | splat_amd.MODE_LIBM_EXP][0], ref)[0] <= 1, desc
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_start_hints_skip_the_scan_and_never_change_a_pixel():
    """SPLAT_OPT_START_HINTS: with a camera at rest the compositor's walks start where the previous frame's did instead of
    scanning for it; with one in slow motion, where they did plus a margin.  The bracket that proves a walk exact is
    closed anew in every frame (and a start that does not do is retried deeper), so every frame -- at rest, creeping,
    stopping, jumping, in libm-exp mode -- is the frame rendered with the hints off, byte for byte; the at-rest frames of
    the default mode stay within 1 LSB of the oracle, the libm-exp ones equal it."""
    from splat_amd import _lib as L
    g = splat_amd.synthetic_scene(90000, 52)
    g.positions[:, :3] *= 0.3                                  # dense: lists beyond 2048 keys, early-out in play
    H, W = 200, 296
    def pose(k):
        # rest (8 frames), creep by 0.2 degrees a frame (12), rest (6), a jump, rest (5), creep (6)
        yaw = 0.3 + np.radians(0.2) * (min(max(k - 8, 0), 12) + max(k - 32, 0)) + (1.0 if k >= 27 else 0.0)
        return make_camera(H, W, (0.0, 0.1, 3.0), yaw=float(yaw))
    poses = [pose(k) for k in range(38)]
    for mode in (0, splat_amd.MODE_LIBM_EXP):
        frames = {}
        for hints in (0, 2, 1):
            r = splat_amd.Renderer(mode=mode) if mode else splat_amd.Renderer()
            try:
                g.compute_cov3d(r)
                r.upload(g)
                r.set_option(L.OPT_START_HINTS, hints)
                imgs = [r.device_image(np.zeros((H, W), np.uint32)) for _ in range(len(poses))]
                for cam, img in zip(poses, imgs):               # asynchronous, several frames in flight
                    r.render_frame_device(cam.to_c(0.01), img)
                r.sync()
                frames[hints] = [r.device_download(img, H, W) for img in imgs]
                st = r.render_frame_device(poses[-1].to_c(0.01), imgs[-1], sync=True, want_stats=True)
                assert st.max_tile_len > 2048
                for img in imgs:
                    r.device_free(img)
            finally:
                r.close()
        for hints in (1, 2):
            for k in range(len(poses)):
                assert np.array_equal(frames[hints][k], frames[0][k]), (mode, hints, k, int((frames[hints][k] != frames[0][k]).sum()))
        for k in (7, 26, 31):                                   # frames at rest, against the oracle
            ref = np.zeros((H, W), np.uint32)
            ref, _ = O.render(scene_dict(g), oracle_camera(poses[k], 0.01), O.default_conventions(), ref, nthreads=8)
            mx, cnt = image_diff(frames[2][k], ref)
            assert (mx == 0) if mode else (mx <= 1), (mode, k, mx, cnt)
