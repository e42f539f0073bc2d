"""GPU tests (-m gpu) of the paths where a frame outgrows the storage sized from earlier frames
(SPLAT_ERR_CAPACITY: tile bucket, pair buffer, sort launch sizes).  A frame that is skipped on the device
leaves its image untouched; a synchronous call redoes its OWN frame, a lost asynchronous frame is reported
once by splat_sync, and the streaming loop (src/main.rs:69-78) redoes it inside splat_stream_wait."""
import os

import numpy as np
import pytest

import splat_amd
from splat_amd import _lib
from splat_amd.renderer import SplatError
from oracle import oracle as O
from helpers import scene_dict, oracle_camera, image_diff, make_camera

pytestmark = pytest.mark.gpu


def oracle_frame(g, cam, init=None):
    h, w = int(cam.h), int(cam.w)
    ref = np.zeros((h, w), np.uint32) if init is None else init.copy()
    ref, ost = O.render(scene_dict(g), oracle_camera(cam, 0.01), O.default_conventions(), ref, nthreads=8)
    return ref, ost


@pytest.fixture()
def scene_and_poses():
    # (regions exactly as their tiles' lists ask: the spare room of the key buffer -- normally handed out, up to four times
    # a region's size -- would let the far pose below fit, and these tests are about frames that do NOT fit)
    saved = os.environ.get("SPLAT_REGION_SPARE")
    os.environ["SPLAT_REGION_SPARE"] = "1"
    try:
        r = splat_amd.Renderer()
    finally:
        os.environ.pop("SPLAT_REGION_SPARE", None)
        if saved is not None:
            os.environ["SPLAT_REGION_SPARE"] = saved
    # (the device-side redo of such frames -- SPLAT_OPT_OVERFLOW_REDO, on by default -- is switched off here: these tests are
    # about what happens when a frame IS skipped; test_overflowing_async_frame_is_binned_again_on_the_device has it on)
    r.set_option(_lib.OPT_OVERFLOW_REDO, 0)
    r.set_option(_lib.OPT_COUNT_FIRST, 0)          # (... and no frame counts its pairs first: one that does cannot outgrow its regions)
    g = splat_amd.synthetic_scene(120000, 71)
    g.compute_cov3d(r)
    near = make_camera(256, 256, (0.0, 0.0, 5.0))
    far = make_camera(256, 256, (0.0, 0.0, 40.0))     # the whole cloud on one or two tiles: lists far beyond their regions
    r.upload(g)
    # every frame slot gets tile regions sized for the NEAR pose (a slot without a layout would count its pairs first and
    # fit any pose: the tests below are about frames that outgrow storage sized from earlier frames)
    warm = np.zeros((256, 256), np.uint32)
    for _ in range(6):
        r.render(near.to_c(0.01), warm)
    yield r, g, near, far
    r.close()


def test_lost_async_frame_is_reported_once_and_never_double_blends(scene_and_poses):
    """an asynchronous frame that overflowed its bucket, followed by a synchronous frame.  The
    synchronous frame composited: it must not be rendered a second time onto its in/out image because of the
    older frame's sticky flag, and the loss must surface at the next splat_sync."""
    r, g, near, far = scene_and_poses
    rng = np.random.default_rng(5)
    init = rng.integers(0, 2**32, (256, 256), dtype=np.uint64).astype(np.uint32)
    a, b, c = r.device_image(init), r.device_image(init), r.device_image(init)
    r.render_device(near.to_c(0.01), a, sync=True)                        # sizes buckets and sort launches for this pose
    assert r.binning_mode() > 0, "one-pass binning expected"
    dropped0 = r.frames_dropped()
    r.render_device(far.to_c(0.01), b, sync=False)                        # outgrows its tile bucket: skipped on the device
    st = r.render_device(near.to_c(0.01), c, sync=True, want_stats=True)  # complete; B's loss is pending
    assert r.frames_dropped() == dropped0 + 1, "the far frame was expected to outgrow its bucket"
    ref_near, ost = oracle_frame(g, near, init)
    got_c = r.device_download(c, 256, 256)
    assert st.n_pairs == ost.n_tile_pairs
    mx, cnt = image_diff(got_c, ref_near)
    assert mx <= 1 and cnt <= 1e-3 * got_c.size, (mx, cnt)                # blended exactly once
    assert np.array_equal(r.device_download(b, 256, 256), init)           # the skipped frame left its image alone
    with pytest.raises(SplatError) as e:
        r.sync()
    assert e.value.code == _lib.ERR_CAPACITY
    r.sync()                                                              # reported once
    # rendered again (synchronously this time) the lost frame is the oracle's
    r.render_device(far.to_c(0.01), b, sync=True)
    ref_far, _ = oracle_frame(g, far, init)
    mx, cnt = image_diff(r.device_download(b, 256, 256), ref_far)
    assert mx <= 1, (mx, cnt)
    for p in (a, b, c):
        r.device_free(p)


def test_synchronous_frame_that_overflows_is_redone_internally(scene_and_poses):
    r, g, near, far = scene_and_poses
    img = np.zeros((256, 256), np.uint32)
    r.render(near.to_c(0.01), img)
    d0 = r.frames_dropped()
    img = np.zeros((256, 256), np.uint32)
    st = r.render(far.to_c(0.01), img)                # host image in and out: overflows, regrown, redone -- no error
    assert r.frames_dropped() > d0
    ref, ost = oracle_frame(g, far)
    assert st.n_pairs == ost.n_tile_pairs
    assert image_diff(img, ref)[0] <= 1
    r.sync()                                          # nothing pending


def test_streamed_frame_that_overflows_is_redone_by_the_wait(scene_and_poses):
    """the viewer loop must survive the first frame whose list outgrows the bucket (C5-class
    scenes, or an orbit whose list-length profile jumps): splat_stream_wait re-renders it."""
    r, g, near, far = scene_and_poses
    f0, f1 = r.host_image(256, 256), r.host_image(256, 256)
    r.render_stream(near.to_c(0.01), f0)
    r.stream_wait(f0)
    d0 = r.frames_dropped()
    r.render_stream(far.to_c(0.01), f1)               # skipped on the device ...
    r.render_stream(near.to_c(0.01), f0)              # ... with another frame queued behind it
    r.stream_wait(f1)                                 # ... and redone here
    r.stream_wait(f0)
    assert r.frames_dropped() > d0
    # the redo settles the loss -- "a viewer loop never sees the miss" -- so nothing may be left
    # pending for the next wait-only entry point (this used to raise a spurious SPLAT_ERR_CAPACITY)
    r.sync()
    r.timing()
    r.sync()
    ref_far, _ = oracle_frame(g, far)
    ref_near, _ = oracle_frame(g, near)
    assert image_diff(np.array(f1), ref_far)[0] <= 1 and np.array(f1).any()
    assert image_diff(np.array(f0), ref_near)[0] <= 1


def test_overflowing_async_frame_is_binned_again_on_the_device(scene_and_poses):
    """Overflow redo (the default): an ASYNCHRONOUS frame whose lists outgrow the regions sized for an earlier camera is
    binned again on the device -- count pass, exact regions, K1, scan, all behind its own scan -- instead of being
    skipped: nothing dropped, nothing reported, the image is the oracle's.  In/out blending onto a non-zero image (a
    frame must blend exactly once), several such frames in flight, and the near pose again afterwards."""
    r, g, near, far = scene_and_poses
    r.set_option(_lib.OPT_OVERFLOW_REDO, 2)                    # on every frame of a moving camera (the default arms itself on the first miss)
    rng = np.random.default_rng(6)
    init = rng.integers(0, 2**32, (256, 256), dtype=np.uint64).astype(np.uint32)
    a, b, c = r.device_image(init), r.device_image(init), r.device_image(init)
    r.render_device(near.to_c(0.01), a, sync=True)
    assert r.binning_mode() > 0, "one-pass binning expected"
    d0 = r.frames_dropped()
    r.render_device(far.to_c(0.01), b, sync=False)            # outgrows its regions: binned again, not skipped
    r.render_device(near.to_c(0.01), c, sync=False)
    r.sync()                                                  # nothing to report
    assert r.frames_dropped() == d0
    ref_far, _ = oracle_frame(g, far, init)
    ref_near, _ = oracle_frame(g, near, init)
    assert image_diff(r.device_download(b, 256, 256), ref_far)[0] <= 1
    assert image_diff(r.device_download(c, 256, 256), ref_near)[0] <= 1
    # cleared frames alternating between the two poses, all in flight
    clear_far, _ = oracle_frame(g, far)
    clear_near, _ = oracle_frame(g, near)
    for k in range(8):
        r.render_frame_device((far if k % 2 else near).to_c(0.01), b if k % 2 else c)
    r.sync()
    assert r.frames_dropped() == d0
    assert image_diff(r.device_download(b, 256, 256), clear_far)[0] <= 1
    assert image_diff(r.device_download(c, 256, 256), clear_near)[0] <= 1
    # the default (adaptive): the launches ride while a list has outgrown its region lately -- and on the frame of a camera JUMP
    # (the view matrix differs by 0.2 or more from the last frame's: this far pose), which used to be the frame that was
    # skipped, reported, and armed the rest.  Quiet for 300 moving frames first (256 disarm; a still camera never counts).
    r.set_option(_lib.OPT_OVERFLOW_REDO, 1)
    wiggle = [near.to_c(0.01), make_camera(256, 256, (0.0, 0.0, 5.001)).to_c(0.01), make_camera(256, 256, (0.0, 0.0, 5.002)).to_c(0.01)]
    for k in range(300):
        r.render_frame_device(wiggle[k % 3], c)
    r.sync()
    d1 = r.frames_dropped()
    r.render_frame_device(far.to_c(0.01), b)
    r.render_frame_device(near.to_c(0.01), c)
    r.render_frame_device(far.to_c(0.01), b)
    r.sync()
    assert r.frames_dropped() == d1
    assert image_diff(r.device_download(b, 256, 256), clear_far)[0] <= 1
    assert image_diff(r.device_download(c, 256, 256), clear_near)[0] <= 1
    for p in (a, b, c):
        r.device_free(p)


def test_regions_that_ask_for_more_than_the_key_buffer_give_up_their_margins():
    """A frame whose lists FIT the key buffer while their regions with the usual margins (1.5 x + 512 keys a tile) do not
    -- here: 16 384 tiles, whose 512-key margins alone exceed the 7.7 M entries (64 N) this small scene starts with: the layout squeezes
    the margins in proportion instead of cutting the last tiles off (build_layout).  Asynchronous frames from the very
    first one, a moving camera with the redo launches on: every frame is rendered; the buffer grows at the first sync."""
    r = splat_amd.Renderer()
    try:
        r.set_option(_lib.OPT_OVERFLOW_REDO, 2)
        g = splat_amd.synthetic_scene(120000, 71)
        g.compute_cov3d(r)
        r.upload(g)
        S, m = 2048, 128 * 128
        cams = [make_camera(S, S, (0.0, 0.0, 5.0 - 0.02 * k)) for k in range(6)]
        img = r.device_image(np.zeros((S, S), np.uint32))
        for cam in cams:
            r.render_frame_device(cam.to_c(0.01), img)
        cap = r.binning_mode()
        r.sync()
        assert r.frames_dropped() == 0, "frames skipped although their lists fit the buffer"
        ref, ost = oracle_frame(g, cams[-1])
        pairs = int(ost.n_tile_pairs)
        assert pairs + 64 * m <= cap < 1.5 * pairs + 500 * m, ("the case this test is about", pairs, cap)
        assert image_diff(r.device_download(img, S, S), ref)[0] <= 1
        st = r.render_frame_device(cams[-1].to_c(0.01), img, sync=True, want_stats=True)
        assert st.n_pairs == pairs
        assert r.binning_mode() > cap, "the buffer grows to the full margins at the sync"
        r.device_free(img)
    finally:
        r.close()


def test_sort_launch_miss_with_tight_grids():
    """the sort launches for long lists (near selection off: with it there are none, and nothing to miss) cover a prefix sized
    from the previous frame; with no margin (SPLAT_DBG_TIGHT_GRIDS) a pose with more long lists than the last one misses,
    is skipped, and redone"""
    from splat_amd import _lib as L
    saved = os.environ.get("SPLAT_DBG_TIGHT_GRIDS")
    os.environ["SPLAT_DBG_TIGHT_GRIDS"] = "1"
    try:
        r = splat_amd.Renderer()
    finally:
        os.environ.pop("SPLAT_DBG_TIGHT_GRIDS", None)
        if saved is not None:
            os.environ["SPLAT_DBG_TIGHT_GRIDS"] = saved
    try:
        g = splat_amd.synthetic_scene(200000, 72)
        g.compute_cov3d(r)
        r.upload(g)
        wide = make_camera(240, 320, (0.0, 0.0, 3.0))       # spread out: few lists reach 2048 keys
        tight = make_camera(240, 320, (0.0, 0.0, 9.0))      # concentrated: many do
        for near in (2048, 0):
            r.set_option(L.OPT_NEAR_SELECT_KEYS, near)
            d0 = r.frames_dropped()
            for cam in (wide, tight, wide, tight):
                img = np.zeros((240, 320), np.uint32)
                st = r.render(cam.to_c(0.01), img)
                ref, ost = oracle_frame(g, cam)
                assert st.n_pairs == ost.n_tile_pairs
                assert image_diff(img, ref)[0] <= 1
            # (near selection is taken only by frames with a list of 8192 keys or a few hundred lists beyond 2048 -- this scene's
            # poses have neither, so both settings size sort launches here; a frame that does select has no launch size to miss:
            # test_near_selection_renders_the_fully_sorted_frame, the full-size tests)
            assert r.frames_dropped() > d0, "expected at least one sort-launch miss"
    finally:
        r.close()


def test_fuzzed_pose_sequences_through_the_frame_pipeline():
    """tools/fuzz_async.py, 15 seeds: random pose sequences rendered asynchronously with several frames in flight
    (storage and launch sizes from earlier frames), slabs against the full frame, streamed frames four in flight --
    every frame equals its synchronous render unless it was skipped on the device, left untouched and reported."""
    import subprocess, sys
    root = os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_async.py"), "15", "4242"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "15 cases, 0 failures" in r.stdout


def test_long_lists_that_outgrow_the_second_key_buffer_grow_it(monkeypatch):
    """The second key buffer of a frame slot holds room for the lists of more than 2048 keys only (their sorted near selection,
    the scatter space of their full sorts), handed out by the frame's scan.  A frame whose long lists ask for more than the
    buffer has is flagged by that scan (overflow 4), skipped, and the buffer grown: a synchronous frame is redone inside its
    call, an asynchronous one is reported once by splat_sync and is right when rendered again."""
    monkeypatch.setenv("SPLAT_DBG_KEYS2_ENTRIES", "4096")
    r = splat_amd.Renderer()
    try:
        g = splat_amd.synthetic_scene(120000, 72)
        g.compute_cov3d(r)
        r.upload(g)
        far = make_camera(256, 256, (0.0, 0.0, 30.0))      # the whole cloud on a few tiles: tens of thousands of keys each
        ref, ost = oracle_frame(g, far)
        d0 = r.frames_dropped()
        img = np.zeros((256, 256), np.uint32)
        st = r.render(far.to_c(0.01), img)
        assert st.max_tile_len > 2048 and st.n_pairs == ost.n_tile_pairs
        assert r.frames_dropped() > d0, "the 4096-entry buffer was expected to be outgrown"
        assert image_diff(img, ref)[0] <= 1
        # the buffer has grown: frames of this pose fit from now on, asynchronous ones included
        d1 = r.frames_dropped()
        dimg = r.device_image(np.zeros((256, 256), np.uint32))
        for _ in range(6):
            r.render_frame_device(far.to_c(0.01), dimg)
        r.sync()
        assert r.frames_dropped() == d1
        assert np.array_equal(r.device_download(dimg, 256, 256), img)
        r.device_free(dimg)
    finally:
        r.close()
    # ... and an ASYNCHRONOUS frame that outgrows it is skipped, reported once, and right when rendered again
    r = splat_amd.Renderer()
    try:
        small = splat_amd.synthetic_scene(2000, 73)              # no list of more than 2048 keys: nothing asked of the buffer yet
        small.compute_cov3d(r)
        r.upload(small)
        warm = np.zeros((256, 256), np.uint32)
        for _ in range(4):
            st = r.render(far.to_c(0.01), warm)
        assert st.max_tile_len <= 2048
        r.upload(g)
        garbage = np.full((256, 256), 0x12345678, np.uint32)
        dimg = r.device_image(garbage)
        d0 = r.frames_dropped()
        r.render_frame_device(far.to_c(0.01), dimg)
        with pytest.raises(SplatError) as e:
            r.sync()
        assert e.value.code == _lib.ERR_CAPACITY
        assert r.frames_dropped() == d0 + 1
        assert np.array_equal(r.device_download(dimg, 256, 256), garbage)      # skipped: untouched
        r.sync()                                                                # reported once
        r.render_frame_device(far.to_c(0.01), dimg, sync=True)
        assert np.array_equal(r.device_download(dimg, 256, 256), img)
        r.device_free(dimg)
    finally:
        r.close()


def test_frames_do_not_depend_on_the_schedule():
    """tools/fuzz_async.py --determinism: every fuzzed pose sequence is rendered by three processes -- the default frame
    pipeline (two binning chains + a compositor in flight), SPLAT_PIPELINE=1 (one stream) and AMD_SERIALIZE_KERNEL=3 (the
    runtime waits around every launch) -- and the frames' digests must agree: bytes that depend on how launches interleave
    are a race (the allocation-time fill race of round 5 would have been caught by this two rounds earlier)."""
    import subprocess, sys
    root = os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_async.py"), "--determinism", "5", "900"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "5 cases x 3 schedules, 0 failures" in r.stdout


def test_first_call_after_allocation_counts_into_zeroed_counters_under_contention():
    """tools/row_loads_stress.py: six processes share the GPU and each asks for the C3 frame's per-tile-row pair counts right
    after its upload, then again between frames.  The context's fills at allocation time are hipMemset calls, which are
    only ENQUEUED (on the legacy default stream) when they return, and the context's streams are non-blocking: with the
    device busy the fill used to land after the first count pass had run (two ranks of eight partitioned the frame from
    counts of 7.4 M and 44 pairs instead of 8 025 623 -- bench.py's gather then died of a size mismatch).  Every answer of
    every process must be the frame's."""
    import subprocess, sys
    root = os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "row_loads_stress.py"), "6", "4"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert "exit codes [0, 0, 0, 0, 0, 0]" in r.stdout
