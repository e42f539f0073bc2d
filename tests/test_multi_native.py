"""The native multi-GPU layer of the C ABI (include/splat_hip.h "Multi-GPU", splat_amd/csrc/splat_multi.hip).

CPU part (-m "not gpu"): splat_slab_partition against the Python partition it mirrors.
GPU part (-m gpu): a frame rendered as slabs by several contexts -- one host thread each, rows gathered to the
root -- must equal the single-context frame BYTE FOR BYTE (SURVEY.md section 8(e) correctness test).  A one-GPU
box can list its device several times (copy transport; RCCL refuses duplicate devices), and can run RCCL
itself with a single rank (ncclCommInitAll / ncclCommInitRank, library loading, partition plumbing)."""
import numpy as np
import pytest

import splat_amd
from splat_amd import dist as sdist
from helpers import make_camera


def test_native_partition_equals_python_partition():
    rng = np.random.default_rng(11)
    for trial in range(200):
        n_rows = int(rng.integers(1, 140))
        k = int(rng.integers(1, 12))
        loads = (rng.random(n_rows) ** 3 * 1e6).astype(np.uint64)
        if trial % 5 == 0:
            loads[rng.integers(0, n_rows, n_rows // 2)] = 0          # empty tile rows
        overhead = float(rng.choice([0.0, 2000.0]))
        want = sdist.slab_partition_balanced(loads, k, row_overhead=overhead)
        got = splat_amd.slab_partition_native(loads, k, overhead)
        assert got == [tuple(s) for s in want], (n_rows, k, got, want)
        # a partition: ordered, disjoint, covering
        assert got[0][0] == 0 and max(b for _, b in got) == n_rows
        for (a0, b0), (a1, b1) in zip(got, got[1:]):
            assert b0 == a1 or (a1 == b1 == n_rows)


def test_native_equal_partition():
    for h, k in ((1080, 8), (2160, 8), (720, 3), (17, 4), (16, 1)):
        want = sdist.slab_partition(h, k)
        got = splat_amd.slab_partition_native(None, k, n_rows=(h + 15) // 16)
        assert got == [tuple(s) for s in want]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0]])
def test_multi_frame_equals_single_context_frame(devices):
    g = splat_amd.synthetic_scene(60000, 33)
    R = splat_amd.Renderer()
    g.compute_cov3d(R)
    cam = make_camera(392, 520, (0.1, 0.0, 4.0), yaw=0.3)      # 25 tile rows, the last one 8 px high
    cam_c = cam.to_c(0.01)
    R.upload(g)
    rng = np.random.default_rng(3)
    init = rng.integers(0, 2**32, (392, 520), dtype=np.uint64).astype(np.uint32)
    single = init.copy()
    st1 = R.render(cam_c, single)
    clear = np.zeros((392, 520), np.uint32)
    R.render(cam_c, clear)
    R.close()

    M = splat_amd.MultiRenderer(devices)
    try:
        M.upload(g)
        slabs = M.balance(cam_c)
        assert slabs[0][0] == 0 and max(b for _, b in slabs) == 25
        # render_to_buffer form: blends onto the caller's host image
        multi = init.copy()
        stm = M.render(cam_c, multi)
        assert np.array_equal(multi, single)
        assert stm.n_pairs == st1.n_pairs            # tile rows are disjoint: pair counts add up exactly
        # viewer-loop form: cleared frames, asynchronous, several in flight
        for _ in range(5):
            M.render_frame(cam_c)
        M.sync()
        assert np.array_equal(M.download(392, 520), clear)
        # ... and with two slab images per rank in turn (splat_multi_set_frame_overlap: consecutive frames composite side
        # by side on every device, the partition weighs rows for that); an odd number of frames: the other image is current
        M.set_frame_overlap(2)
        for _ in range(5):
            M.render_frame(cam_c)
        M.sync()
        assert np.array_equal(M.download(392, 520), clear)
        M.render_frame(cam_c)
        M.sync()
        assert np.array_equal(M.download(392, 520), clear)
        multi = init.copy()
        M.render(cam_c, multi)                       # the host in/out form between overlapped frames
        assert np.array_equal(multi, single)
        M.set_frame_overlap(1)
        # another target size re-partitions by itself
        cam2 = make_camera(200, 300)
        img2 = np.zeros((200, 300), np.uint32)
        M.render(cam2.to_c(0.01), img2)
        R2 = splat_amd.Renderer()
        try:
            R2.upload(g)
            ref2 = np.zeros((200, 300), np.uint32)
            R2.render(cam2.to_c(0.01), ref2)
        finally:
            R2.close()
        assert np.array_equal(img2, ref2)
    finally:
        M.close()


@pytest.mark.gpu
def test_comm_single_rank_over_rccl():
    """form (A) with one rank: RCCL is loaded, ncclGetUniqueId / ncclCommInitRank run on the GPU box, the partition
    is applied, and a gather with nothing to exchange leaves the frame alone"""
    g = splat_amd.synthetic_scene(20000, 34)
    R = splat_amd.Renderer()
    try:
        g.compute_cov3d(R)
        R.upload(g)
        cam_c = make_camera(128, 192).to_c(0.01)
        ref = np.zeros((128, 192), np.uint32)
        R.render(cam_c, ref)
        uid = splat_amd.Renderer.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        R.comm_init(uid, 1, 0)
        R.comm_set_slabs(splat_amd.slab_partition_native(R.tile_row_loads(cam_c), 1, 2000.0))
        d = R.device_image(np.zeros((128, 192), np.uint32))
        R.render_device(cam_c, d)
        R.comm_gather(d, 192, 128, 0)
        R.sync()
        assert np.array_equal(R.device_download(d, 128, 192), ref)
        R.device_free(d)
    finally:
        R.close()


@pytest.mark.gpu
def test_rccl_send_recv_really_runs_in_the_single_rank_loopback():
    """with one rank the gather returned before ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd
    ever ran, on any hardware.  The loopback hook sends this rank's slab rows through RCCL to itself (send + recv to the
    own rank in one group, on the context's stream), overwrites the image rows with a marker and restores them from
    what RCCL delivered: the frame must come back byte for byte -- and the marker must NOT (it would if the receive
    had delivered nothing).  Run under rocprofv3 --kernel-trace the log shows RCCL's kernel (profiles/r04_rccl_loopback*)."""
    g = splat_amd.synthetic_scene(20000, 35)
    R = splat_amd.Renderer()
    try:
        g.compute_cov3d(R)
        R.upload(g)
        cam_c = make_camera(200, 320).to_c(0.01)          # 13 tile rows, the last one 8 px high
        rng = np.random.default_rng(9)
        init = rng.integers(0, 2**32, (200, 320), dtype=np.uint64).astype(np.uint32)
        ref = init.copy()
        R.render(cam_c, ref)
        R.comm_init(splat_amd.Renderer.comm_unique_id(), 1, 0)
        R.comm_set_slabs([(0, 13)])
        with pytest.raises(splat_amd.SplatError):
            R.comm_gather(0, 320, 200, 0)                 # NULL image: argument check comes first
        R.comm_loopback(True)
        d = R.device_image(init)
        R.render_device(cam_c, d)
        R.comm_gather(d, 320, 200, 0)                     # ncclGroupStart, ncclSend, ncclRecv, ncclGroupEnd on the context's stream
        R.sync()
        got = R.device_download(d, 200, 320)
        assert np.array_equal(got, ref)
        assert not (got == 0x5a5a5a5a).all()
        # a thinner slab: only its rows travel
        R.comm_set_slabs([(3, 7)])
        R.render_device(cam_c, d)                         # blends the slab's rows a second time: compare with the same on the host
        R.comm_gather(d, 320, 200, 0)
        R.sync()
        ref2 = ref.copy()
        R.comm_loopback(False)
        d2 = R.device_image(ref)
        R.render_device(cam_c, d2)
        R.comm_gather(d2, 320, 200, 0)                    # single rank, hook off: nothing to exchange
        R.sync()
        assert np.array_equal(R.device_download(d, 200, 320), R.device_download(d2, 200, 320))
        assert np.array_equal(ref2[:48], R.device_download(d2, 200, 320)[:48])      # rows outside the slab untouched
        R.device_free(d); R.device_free(d2)
    finally:
        R.close()


@pytest.mark.gpu
def test_fuzzed_partitions_and_pose_sequences_equal_the_single_context():
    """tools/fuzz_multi.py, 12 seeds: random scenes, target sizes, 1..6 ranks sharing this GPU, balanced and equal slabs,
    host in/out frames onto a random image, viewer-loop frames queued back to back over changing poses (a slab skipped
    for capacity is reported by the sync and redone), re-balancing: the gathered frame is the single-context frame."""
    import os, subprocess, sys
    root = os.path.join(os.path.dirname(__file__), "..")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_multi.py"), "12", "777"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "12 cases, 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
