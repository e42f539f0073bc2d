"""BASELINE.json configs 4 and 5 at WORKLOAD size (-m gpu): the C3 frame (1.5 M Gaussians @1920x1080) as 2 / 4 / 8
load-balanced tile-row slabs and the C5 frame (6 M @3840x2160) as 8, every slab rendered by a context of its own (one
host thread each, splat_multi_*: form (B) of include/splat_hip.h "Multi-GPU") and the slab rows gathered into the root's
image -- the gathered frame must be the single-context frame BYTE FOR BYTE (pixels are
independent given the ordered splat list, src/pipelines.rs:147-168).

One GPU carries all the ranks here (a device listed k times: rows travel as device copies; RCCL refuses duplicate
devices) -- the decomposition, the balanced partition, the per-slab block culling and the region layouts of every slab
are the ones a node of k GPUs runs; only the transport differs (covered by the RCCL loopback test and bench.py --gpus N).
What it stands for in the reference: one render_to_buffer call (src/pipelines.rs:66-86) whose rows euc fans out to
CPU threads."""
import numpy as np
import pytest

import splat_amd
from bench import WORKLOADS, make_scene

pytestmark = pytest.mark.gpu

_cache = {}


def workload(name):
    """scene + the single-context frames (cleared and blended-onto-noise) of a workload, kept across its cases"""
    if _cache.get("name") != name:
        _cache.clear()
        n, W, H, seed = WORKLOADS[name]
        R = splat_amd.Renderer()
        try:
            g = make_scene(name)
            g.compute_cov3d(R)
            R.upload(g)
            cam = splat_amd.Camera(H, W, (0.0, 0.0, 5.0))
            cam.update_camera_pose()
            cam_c = cam.to_c(0.01, 15)
            clear = np.zeros((H, W), np.uint32)
            st = R.render(cam_c, clear)
            # a second pose of the orbit (src/main.rs:53-60): the partition is re-balanced for it
            cam.update_yaw_angle(np.radians(70.0))
            cam.update_camera_pose()
            cam2_c = cam.to_c(0.01, 15)
            clear2 = np.zeros((H, W), np.uint32)
            R.render(cam2_c, clear2)
        finally:
            R.close()
        _cache.update(name=name, g=g, W=W, H=H, cam_c=cam_c, cam2_c=cam2_c, clear=clear, clear2=clear2, n_pairs=int(st.n_pairs))
    return _cache


@pytest.fixture(scope="module", autouse=True)
def _release():
    yield
    _cache.clear()


@pytest.mark.parametrize("wl,k", [("C3", 2), ("C3", 4), ("C3", 8), ("C5", 8)])
def test_fullsize_frame_as_slabs_equals_single_context_frame(wl, k):
    c = workload(wl)
    W, H = c["W"], c["H"]
    tile_rows = (H + 15) // 16
    M = splat_amd.MultiRenderer([0] * k)
    try:
        M.upload(c["g"])
        slabs = M.balance(c["cam_c"])
        assert len(slabs) == k and slabs[0][0] == 0 and max(b for _, b in slabs) == tile_rows
        assert all(b > a for a, b in slabs), slabs                 # every rank has rows at this size
        # render_to_buffer form (host in/out), with statistics: the slabs' pair counts add up to the frame's
        img = np.zeros((H, W), np.uint32)
        st = M.render(c["cam_c"], img)
        assert st.n_pairs == c["n_pairs"], (st.n_pairs, c["n_pairs"])
        assert np.array_equal(img, c["clear"]), int((img != c["clear"]).sum())
        # viewer-loop form: asynchronous cleared frames, several in flight, one image per rank ...
        for _ in range(4):
            M.render_frame(c["cam_c"])
        M.sync()
        assert np.array_equal(M.download(H, W), c["clear"])
        # ... and two images per rank in turn (the swap chain bench.py's ranks use over RCCL)
        M.set_frame_overlap(2)
        for _ in range(5):
            M.render_frame(c["cam_c"])
        M.sync()
        assert np.array_equal(M.download(H, W), c["clear"])
        # another pose: re-balanced slabs (regions and culling verdicts of every slab are rebuilt)
        slabs2 = M.balance(c["cam2_c"])
        assert max(b for _, b in slabs2) == tile_rows
        for _ in range(3):
            M.render_frame(c["cam2_c"])
        M.sync()
        assert np.array_equal(M.download(H, W), c["clear2"])
        assert sum(M.rank_frames_dropped(r) for r in range(k)) == 0
    finally:
        M.close()
