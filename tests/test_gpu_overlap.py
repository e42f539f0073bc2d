"""GPU tests (-m gpu) of the compositor lanes (splat_set_frame_overlap, include/splat_hip.h): asynchronous frames to
DIFFERENT device images may composite side by side; frames to the SAME image keep their call order (the in/out blend of
render_to_buffer, src/gaussians.rs:359-372, and the viewer loop's clear + render, src/main.rs:71-75), synchronous frames
follow everything in flight, the gather follows the frame rendered last.  Every frame is the serial frame bit for bit."""
import numpy as np
import pytest

import splat_amd
from splat_amd import _lib
from helpers import make_camera

pytestmark = pytest.mark.gpu

H, W = 480, 640


@pytest.fixture(scope="module")
def scene():
    g = splat_amd.synthetic_scene(150000, 91)
    r = splat_amd.Renderer()
    g.compute_cov3d(r)
    r.close()
    return g


def poses():
    # (neighbouring poses: every frame is another image, and none outgrows the tile regions sized from the frames before it)
    return [make_camera(H, W, (0.02 * k, 0.01 * k, 4.0 + 0.03 * k), yaw=0.01 * k).to_c(0.01) for k in range(6)]


def settled(r):
    """splat_sync; False when an asynchronous frame was skipped on the device (reported once, SPLAT_ERR_CAPACITY)"""
    try:
        r.sync()
        return True
    except splat_amd.renderer.SplatError as e:
        assert e.code == _lib.ERR_CAPACITY
        return False


def serial_frames(g, cams, init, clear):
    """every pose rendered synchronously by a context with the default (no overlap)"""
    r = splat_amd.Renderer()
    try:
        r.upload(g)
        out = []
        for c in cams:
            d = r.device_image(init)
            (r.render_frame_device if clear else r.render_device)(c, d, sync=True)
            out.append(r.device_download(d, H, W))
            r.device_free(d)
        return out
    finally:
        r.close()


def test_swap_chain_frames_are_the_serial_frames(scene):
    """two images in turn, frames enqueued back to back: lane 0, lane 1, lane 0 ... -- each frame the serial frame"""
    cams = poses()
    rng = np.random.default_rng(3)
    init = rng.integers(0, 2**32, (H, W), dtype=np.uint64).astype(np.uint32)
    want = serial_frames(scene, cams, init, clear=True)
    r = splat_amd.Renderer()
    try:
        r.upload(scene)
        r.set_frame_overlap(2)
        imgs = [r.device_image(init), r.device_image(init)]
        for c in cams:
            r.render_frame_device(c, imgs[0], sync=True)
        checked = 0
        for rep in range(6):
            for k0 in range(0, len(cams), 2):
                for k in (k0, k0 + 1):
                    r.render_frame_device(cams[k], imgs[k % 2])              # asynchronous: the pair shares the chip
                if not settled(r):
                    continue
                for k in (k0, k0 + 1):
                    assert np.array_equal(r.device_download(imgs[k % 2], H, W), want[k]), (rep, k)
                checked += 1
        assert checked >= 12, checked
        for d in imgs:
            r.device_free(d)
    finally:
        r.close()


def test_frames_to_one_image_keep_their_order_with_overlap_on(scene):
    """in/out blending (splat_render_device blends onto what the image holds): X, X, Y, Y, X, then a synchronous frame onto Y.
    Each image must hold its frames blended in call order -- a frame follows the earlier frames to ITS image on their lane
    (the pattern is not the lanes' own alternation: a library that alternated blindly would let the two frames of a pair
    blend onto one image at the same time; checked with such a build)"""
    cams = poses()
    init = np.full((H, W), 0xff204060, np.uint32)
    r0 = splat_amd.Renderer()
    r0.upload(scene)
    x = r0.device_image(init); y = r0.device_image(init)
    for c, d in ((cams[0], x), (cams[1], x), (cams[2], y), (cams[3], y), (cams[4], x), (cams[5], y)):
        r0.render_device(c, d, sync=True)
    want_x, want_y = r0.device_download(x, H, W), r0.device_download(y, H, W)
    r0.close()
    r = splat_amd.Renderer()
    try:
        r.upload(scene)
        r.set_frame_overlap(2)
        for c in cams:                                       # regions and launch sizes for these poses, so that nothing is skipped below
            warm = r.device_image(init); r.render_device(c, warm, sync=True); r.device_free(warm)
        for rep in range(8):
            x = r.device_image(init); y = r.device_image(init)
            d0 = r.frames_dropped()
            r.render_device(cams[0], x)
            r.render_device(cams[1], x)
            r.render_device(cams[2], y)
            r.render_device(cams[3], y)
            r.render_device(cams[4], x)
            r.render_device(cams[5], y, sync=True)           # a synchronous frame follows everything in flight, on both lanes
            got_y = r.device_download(y, H, W)
            got_x = r.device_download(x, H, W)
            settled(r)
            if r.frames_dropped() == d0:                     # (a skipped asynchronous frame is reported, not redone: not this test's subject)
                assert np.array_equal(got_x, want_x), rep
                assert np.array_equal(got_y, want_y), rep
            r.device_free(x); r.device_free(y)
    finally:
        r.close()


def test_three_images_in_rotation_with_a_long_frame_in_front(scene):
    """triple buffering: X (from inside the cloud: a long binning chain, so its compositor starts late), W (from far away:
    the cloud on a few tiles, a long compositor), Y, X again.  The second frame to X must not start before the first has
    finished although two other frames were enqueued in between -- the hazard is decided per image, not from each lane's
    last frame only (a build that did the latter left the first frame's pixels in tens of thousands of places: the empty
    tiles of the second frame are written at once, the first frame's tail lands on top).  Inside and far poses alternate, so
    that every frame's tile regions come from a frame of its own kind two frames earlier and none is skipped."""
    seq = [make_camera(H, W, (0.1, 0.1, 0.6), yaw=0.4).to_c(0.01), make_camera(H, W, (0.0, 0.0, 9.0)).to_c(0.01),
           make_camera(H, W, (0.11, 0.1, 0.61), yaw=0.4).to_c(0.01), make_camera(H, W, (0.02, 0.0, 9.05)).to_c(0.01)]
    init = np.zeros((H, W), np.uint32)
    scene = splat_amd.synthetic_scene(700000, 92)           # (long lists from inside: the first frame's compositor outlasts two whole short frames)
    r = splat_amd.Renderer()
    scene.compute_cov3d(r)
    r.close()
    want = serial_frames(scene, seq, init, clear=True)
    r = splat_amd.Renderer()
    try:
        r.upload(scene)
        r.set_frame_overlap(2)
        x, w, y = r.device_image(init), r.device_image(init), r.device_image(init)
        checked = 0
        for rep in range(8):
            for c in seq * 2:                                 # (synchronous frames: every repetition starts from the same lanes)
                r.render_frame_device(c, x, sync=True)
            r.render_frame_device(seq[0], x)
            r.render_frame_device(seq[1], w)
            r.render_frame_device(seq[2], y)
            r.render_frame_device(seq[3], x)
            if not settled(r):
                continue
            assert np.array_equal(r.device_download(x, H, W), want[3]), rep
            assert np.array_equal(r.device_download(w, H, W), want[1]), rep
            assert np.array_equal(r.device_download(y, H, W), want[2]), rep
            checked += 1
        assert checked >= 4, checked
        for d in (x, w, y):
            r.device_free(d)
    finally:
        r.close()


def test_overlap_setting_is_validated_and_can_be_switched_back(scene):
    r = splat_amd.Renderer()
    try:
        with pytest.raises(splat_amd.renderer.SplatError):
            r.set_frame_overlap(3)
        with pytest.raises(splat_amd.renderer.SplatError):
            r.set_frame_overlap(0)
        r.upload(scene)
        cam = poses()[0]
        init = np.zeros((H, W), np.uint32)
        a, b = r.device_image(init), r.device_image(init)
        r.set_frame_overlap(2)
        r.render_frame_device(cam, a); r.render_frame_device(cam, b)
        r.set_frame_overlap(1)                               # waits for what is in flight
        r.render_frame_device(cam, a); r.render_frame_device(cam, b)
        r.sync()
        assert np.array_equal(r.device_download(a, H, W), r.device_download(b, H, W))
        assert r.device_download(a, H, W).any()
    finally:
        r.close()


def test_gather_follows_its_frame_on_either_lane(scene):
    """a single-rank communicator in loopback mode (the slab rows really travel through ncclSend + ncclRecv): frame + gather
    to two images in turn with overlap on -- each image afterwards holds its own frame"""
    cams = poses()
    init = np.zeros((H, W), np.uint32)
    want = serial_frames(scene, cams[:4], init, clear=True)
    r = splat_amd.Renderer()
    try:
        r.upload(scene)
        try:
            r.comm_init(splat_amd.Renderer.comm_unique_id(), 1, 0)
        except splat_amd.renderer.SplatError as e:
            pytest.skip("RCCL unavailable: %s" % e)
        r.comm_set_slabs([(0, (H + 15) // 16)])
        r.comm_loopback(True)
        r.set_frame_overlap(2)
        imgs = [r.device_image(init), r.device_image(init)]
        for c in cams[:4]:
            r.render_frame_device(c, imgs[0], sync=True)
        for rep in range(3):
            for k0 in (0, 2):
                for k in (k0, k0 + 1):
                    r.render_frame_device(cams[k], imgs[k % 2])
                    r.comm_gather(imgs[k % 2], W, H, 0)
                if not settled(r):
                    continue
                for k in (k0, k0 + 1):
                    assert np.array_equal(r.device_download(imgs[k % 2], H, W), want[k]), (rep, k)
        # ... and a download right behind a gather, with no sync in between (the copy entry points follow the
        # frame's "ended" event, which the gather now re-records behind itself -- on whichever lane the frame ran)
        for k in (0, 1, 2, 3):
            r.render_frame_device(cams[k], imgs[k % 2])
            r.comm_gather(imgs[k % 2], W, H, 0)
            got = r.device_download(imgs[k % 2], H, W)
            if r.frames_dropped() == 0:
                assert np.array_equal(got, want[k]), k
        settled(r)
        for d in imgs:
            r.device_free(d)
    finally:
        r.close()


def test_render_into_a_registered_host_image(scene):
    """splat_host_register: the caller's own (pageable) image page-locked once -- render_to_buffer's host in/out form
    (src/gaussians.rs:359-372) gives the same frame, and the registration can be dropped and made again"""
    cam = poses()[0]
    rng = np.random.default_rng(11)
    init = rng.integers(0, 2**32, (H, W), dtype=np.uint64).astype(np.uint32)
    r = splat_amd.Renderer()
    try:
        r.upload(scene)
        plain = init.copy()
        r.render(cam, plain)
        pinned = init.copy()
        splat_amd.Renderer.host_register(pinned)
        try:
            r.render(cam, pinned)
            assert np.array_equal(pinned, plain)
        finally:
            splat_amd.Renderer.host_unregister(pinned)
        splat_amd.Renderer.host_register(pinned)
        splat_amd.Renderer.host_unregister(pinned)
    finally:
        r.close()
