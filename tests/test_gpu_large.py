"""GPU tests (-m gpu) of the LARGE-SPLAT path of one-pass binning (round 7): a splat of more than SPLAT_LARGE_TILES tiles, or
wider / taller than K1's 32 x 32-tile window, is not expanded by its K1 block; it goes to the frame's large list and
bin_large_kernel bins the list tile by tile behind K1.  What replaces what: the instance expansion of src/pipelines.rs:69-79
and euc's row loops (src/pipelines.rs:80-84) for the splats that fill the screen -- a camera inside the scene.
The lists must be the oracle's lists whatever the threshold, through every chain a frame can take (plain, count first,
binned again on the device, slab), and byte for byte the frames of the path without a list."""
import os

import numpy as np
import pytest

import splat_amd
import splat_amd.renderer
from splat_amd import _lib
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, image_diff, make_camera

pytestmark = pytest.mark.gpu


def renderer(large_tiles=None, **env):
    keys = dict(env)
    if large_tiles is not None:
        keys["SPLAT_LARGE_TILES"] = str(large_tiles)
    saved = {k: os.environ.get(k) for k in keys}
    os.environ.update({k: str(v) for k, v in keys.items()})
    try:
        return splat_amd.Renderer()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def closeup_scene(n=60000, seed=77):
    """a cloud tight around the origin with a few hundred very large Gaussians mixed in: from inside, hundreds of splats cover
    hundreds to all of the tiles"""
    g = splat_amd.synthetic_scene(n, seed)
    g.positions[:, :3] *= 0.5
    rng = np.random.default_rng(seed)
    big = rng.choice(n, 400, replace=False)
    g.scales[big] *= rng.uniform(4.0, 30.0, (400, 1)).astype(np.float32)
    return g


POSES = [((0.0, 0.0, 5.0), 0.0, 0.0), ((0.05, 0.1, 0.3), 0.4, 0.0), ((0.2, -0.1, 0.0), 2.0, 0.3), ((0.0, 0.0, 1.2), 0.0, -0.2)]


def expected_lists(g, cam, tiles_x, tiles_y):
    want = O.preprocess(scene_dict(g), oracle_camera(cam, 0.01))
    glob = O.sort(g.positions, np.array(cam.to_c(0.01).view[:], np.float32))
    rank = np.empty(len(g), np.int64)
    rank[glob] = np.arange(len(g))
    v = want["visible"] == 1
    tx0, tx1, ty0, ty1 = want["px0"] // 16, want["px1"] // 16, want["py0"] // 16, want["py1"] // 16
    ntiles = np.where(v, (tx1 - tx0 + 1) * (ty1 - ty0 + 1), 0)
    lists = []
    for t in range(tiles_x * tiles_y):
        tx, ty = t % tiles_x, t // tiles_x
        exp = np.nonzero(v & (tx0 <= tx) & (tx1 >= tx) & (ty0 <= ty) & (ty1 >= ty))[0]
        lists.append(exp[np.argsort(rank[exp], kind="stable")])
    return lists, ntiles


@pytest.mark.parametrize("large_tiles", [None, 0, 1, 16, 1000])
def test_tile_lists_and_frames_are_the_oracles_for_any_threshold(large_tiles):
    """the default, 0 (only the window decides), 1 (every splat of two tiles or more goes through the list), 16, 1000"""
    r = renderer(large_tiles)
    try:
        g = closeup_scene()
        g.compute_cov3d(r)
        r.upload(g)
        h, w = 272, 400                    # 25 x 17 tiles: neither a multiple of the 4 x 4 tile groups
        tiles_x, tiles_y = 25, 17
        for pos, yaw, pitch in POSES:
            cam = make_camera(h, w, pos, yaw, pitch)
            img = np.zeros((h, w), np.uint32)
            st = r.render(cam.to_c(0.01), img)
            assert r.binning_mode() > 0, "one-pass binning expected"
            lists, ntiles = expected_lists(g, cam, tiles_x, tiles_y)
            assert st.n_pairs == sum(len(x) for x in lists), (pos, st.n_pairs)
            if pos != POSES[0][0]:
                assert (ntiles > 128).sum() > 50, "the pose is meant to have large splats"
            off, order = r.tile_lists(tiles_x * tiles_y, st.n_pairs)
            for t, exp in enumerate(lists):
                assert np.array_equal(order[off[t]:off[t + 1]], exp), ("tile", t, pos)
            ref, ost = O.render(scene_dict(g), oracle_camera(cam, 0.01), O.default_conventions(), np.zeros((h, w), np.uint32), nthreads=8)
            mx, cnt = image_diff(img, ref)
            assert st.n_pairs == ost.n_tile_pairs and mx <= 1 and cnt <= 1e-3 * h * w, (pos, mx, cnt)
    finally:
        r.close()


def test_frames_are_byte_identical_with_and_without_the_list_through_every_chain():
    """SPLAT_LARGE_TILES=-1 is round 5's path (K1's blocks expand their close-ups themselves).  A pose sequence that jumps into
    the cloud and out again with frames in flight -- first frames (count first), frames binned into another camera's regions,
    frames binned again on the device (overflow redo forced on), a slab -- gives the same bytes and loses no frame."""
    g = closeup_scene(80000, 78)
    h, w = 360, 640
    seq = [POSES[0], POSES[0], POSES[1], POSES[1], POSES[2], POSES[0], POSES[3], POSES[3], POSES[1], POSES[0], POSES[2], POSES[2]]
    out = {}
    # (-1: no list ever; default; 1: every splat of two tiles or more through the list; "lazy": the default threshold but a list only
    # after a frame with a million large splats, i.e. on the first frames and behind jumps -- the policy's on / off switch)
    for lt in (-1, None, 1, "lazy"):
        for redo in (1, 2):
            r = renderer(None, SPLAT_REGION_SPARE=1, SPLAT_LARGE_LIST_MIN=1000000) if lt == "lazy" else renderer(lt, SPLAT_REGION_SPARE=1)
            try:
                r.set_option(_lib.OPT_OVERFLOW_REDO, redo)
                g.compute_cov3d(r)
                r.upload(g)
                imgs = [r.device_image(np.zeros((h, w), np.uint32)) for _ in seq]
                d0 = r.frames_dropped()
                for k, (pos, yaw, pitch) in enumerate(seq):
                    r.render_frame_device(make_camera(h, w, pos, yaw, pitch).to_c(0.01), imgs[k])
                try:
                    r.sync()
                except splat_amd.renderer.SplatError as e:            # (adaptive redo: the first frame that outgrows a region after a
                    assert redo == 1 and e.code == _lib.ERR_CAPACITY  # quiet stretch is skipped, reported once, and arms the redo)
                frames = [r.device_download(p, h, w) for p in imgs]
                assert r.frames_dropped() == d0 or redo == 1
                if r.frames_dropped() != d0:
                    frames = None
                # a slab of the last pose
                r.set_slab(5, 14)
                slab = np.zeros((h, w), np.uint32)
                r.render_frame(make_camera(h, w, *seq[-1]).to_c(0.01), slab)
                r.set_slab(0, -1)
                out[(lt, redo)] = (frames, slab)
                for p in imgs:
                    r.device_free(p)
            finally:
                r.close()
    ref_frames, ref_slab = out[(-1, 2)]
    assert ref_frames is not None
    assert any(f.any() for f in ref_frames)
    for key, (frames, slab) in out.items():
        assert np.array_equal(slab, ref_slab), key
        if frames is not None:
            for k, (a, b) in enumerate(zip(frames, ref_frames)):
                assert np.array_equal(a, b), (key, k)
    assert out[(None, 2)][0] is not None and out[(1, 2)][0] is not None and out[("lazy", 2)][0] is not None


def test_count_only_pass_counts_the_large_splats():
    """splat_tile_row_loads (the slab balancer's count pass) and the first frame of a context (count first) agree with the
    frame's pair count when most pairs come from large splats"""
    r = renderer()
    try:
        g = closeup_scene(30000, 79)
        g.compute_cov3d(r)
        r.upload(g)
        h, w = 256, 384
        cam = make_camera(h, w, *POSES[1])
        img = np.zeros((h, w), np.uint32)
        st = r.render(cam.to_c(0.01), img)               # the context's first frame: every slot counts first
        lists, ntiles = expected_lists(g, cam, 24, 16)
        assert st.n_pairs == sum(len(x) for x in lists)
        assert ntiles[ntiles > 128].sum() > 0.3 * st.n_pairs
        loads = r.tile_row_loads(cam.to_c(0.01))
        assert int(np.sum(loads)) == st.n_pairs
        for _ in range(5):                                # ... and the frames behind it (regions sized from the lists) still do
            st2 = r.render(cam.to_c(0.01), np.zeros((h, w), np.uint32))
            assert st2.n_pairs == st.n_pairs
        assert r.frames_dropped() == 0
    finally:
        r.close()


def test_motion_sized_regions_change_no_pixel_and_lose_no_frame():
    """A moving camera's tile regions are sized from the longest list AROUND each tile (build_layout's motion filter,
    SPLAT_LAYOUT_MOTION; the radius is the frame policy's layout_radius).  Where the keys land in the buffer is all that changes:
    an asynchronous yaw path at 2 and 8 degrees a frame gives the same bytes with the filter on and off, frame for frame, and
    with the overflow redo on every moving frame nothing is dropped either way."""
    g = closeup_scene(60000, 80)
    h, w = 368, 640
    out = {}
    for lm in (0, 1):
        r = renderer(None, SPLAT_LAYOUT_MOTION=lm, SPLAT_REGION_SPARE=1)
        try:
            r.set_option(_lib.OPT_OVERFLOW_REDO, 2)
            g.compute_cov3d(r)
            r.upload(g)
            cam = splat_amd.Camera(h, w, (0.0, 0.0, 2.2))
            poses = []
            for k in range(40):
                cam.update_camera_pose()
                poses.append(cam.to_c(0.01))
                cam.update_yaw_angle(np.radians(2.0 if k < 20 else 8.0))
            imgs = [r.device_image(np.zeros((h, w), np.uint32)) for _ in poses]
            d0 = r.frames_dropped()
            for c, p in zip(poses, imgs):
                r.render_frame_device(c, p)
            r.sync()
            assert r.frames_dropped() == d0
            out[lm] = [r.device_download(p, h, w) for p in imgs]
            for p in imgs:
                r.device_free(p)
        finally:
            r.close()
    assert any(f.any() for f in out[0])
    for k, (a, b) in enumerate(zip(out[0], out[1])):
        assert np.array_equal(a, b), k
    # ... and the frames are the oracle's (every tenth pose)
    cam = splat_amd.Camera(h, w, (0.0, 0.0, 2.2))
    for k in range(40):
        cam.update_camera_pose()
        if k % 10 == 0:
            ref, _ = O.render(scene_dict(g), oracle_camera(cam, 0.01), O.default_conventions(), np.zeros((h, w), np.uint32), nthreads=8)
            mx, cnt = image_diff(out[1][k], ref)
            assert mx <= 1 and cnt <= 1e-3 * h * w, (k, mx, cnt)
        cam.update_yaw_angle(np.radians(2.0 if k < 20 else 8.0))
