"""N>1 path on CPU: world_size-2 gloo processes, each fills the rows of its tile-row slab (with the
oracle standing in for the GPU kernels, which cannot run here), then the product's gather_slabs()
assembles the frame on rank 0.  Must equal the single-process frame byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import splat_amd
    from splat_amd import dist as sdist
    from oracle import oracle as O
    from helpers import scene_dict, oracle_camera, make_camera, with_oracle_cov3d
    g = with_oracle_cov3d(splat_amd.synthetic_scene(3000, 23))
    cam = make_camera(h, w)
    slabs = sdist.slab_partition(h, world)
    r0, r1 = sdist.slab_pixel_rows(slabs[rank], h)
    img = np.zeros((h, w), np.uint32)
    O.render(scene_dict(g), oracle_camera(cam, 0.01), argb=img, rows=(r0, r1))
    t = torch.from_numpy(img.view(np.int32))
    sdist.gather_slabs(t, slabs, rank)
    if rank == 0:
        np.save(out_path, t.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,w", [(2, 200, 160), (3, 72, 96)])
def test_gather_slabs_gloo(tmp_path, world, h, w):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import splat_amd
    from oracle import oracle as O
    from helpers import scene_dict, oracle_camera, make_camera, with_oracle_cov3d
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), h, w, out), nprocs=world, join=True)
    got = np.load(out)
    g = with_oracle_cov3d(splat_amd.synthetic_scene(3000, 23))
    want, _ = O.render(scene_dict(g), oracle_camera(make_camera(h, w), 0.01))
    assert np.array_equal(got, want)


def test_slab_partition_covers_all_tile_rows():
    from splat_amd.dist import slab_partition, slab_pixel_rows
    for h in (1, 15, 16, 17, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            slabs = slab_partition(h, world)
            assert len(slabs) == world and slabs[0][0] == 0
            assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
            assert slabs[-1][1] == (h + 15) // 16
            rows = [slab_pixel_rows(s, h) for s in slabs]
            assert rows[-1][1] == h and sum(b - a for a, b in rows) == h
    # 1080p on 8 GPUs -> 9,9,9,9,8,8,8,8 tile rows
    assert [b - a for a, b in slab_partition(1080, 8)] == [9, 9, 9, 9, 8, 8, 8, 8]
    assert [b - a for a, b in slab_partition(2160, 8)] == [17, 17, 17, 17, 17, 17, 17, 16]


def test_balanced_partition_properties():
    from splat_amd.dist import slab_partition_balanced
    rng = np.random.default_rng(0)
    for n in (1, 5, 68, 135):
        for world in (1, 2, 4, 8):
            loads = np.exp(-0.5 * ((np.arange(n) - n / 2) / (n / 6 + 1)) ** 2) * 1e5 + rng.integers(0, 100, n)
            slabs = slab_partition_balanced(loads, world)
            assert len(slabs) == world
            assert slabs[0][0] == 0 and max(b for _, b in slabs) == n
            assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]) if b[1] > b[0])
            covered = sum(b - a for a, b in slabs)
            assert covered == n
            if world <= n:
                assert all(b > a for a, b in slabs)
            # balanced beats (or ties) the equal-rows split on the bottleneck
            eq = [(n * k // world, n * (k + 1) // world) for k in range(world)]
            worst = lambda ss: max(loads[a:b].sum() if b > a else 0 for a, b in ss)
            assert worst(slabs) <= worst(eq) * 1.0001 + 1e-9
