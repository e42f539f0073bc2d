"""Multi-GPU decomposition of a frame: disjoint tile-row slabs, one process per GPU, one gather of
slab pixels to rank 0 per frame (RCCL send/recv group over xGMI; gloo on CPU for tests).

Pixels are independent given the globally ordered splat list (blend state is per pixel,
src/pipelines.rs:147-168), so every rank keeps the whole scene, bins only the tiles of its slab and
the only exchange is the final image rows -- SURVEY.md section 8(e)."""
from ._lib import TILE


def slab_partition(height, world_size):
    """Tile rows [r0, r1) per rank: ceil-balanced, earlier ranks take the extra row."""
    tiles_y = (int(height) + TILE - 1) // TILE
    base, extra = divmod(tiles_y, world_size)
    out, r = [], 0
    for k in range(world_size):
        n = base + (1 if k < extra else 0)
        out.append((r, r + n))
        r += n
    return out


def slab_pixel_rows(slab, height):
    return min(slab[0] * TILE, int(height)), min(slab[1] * TILE, int(height))


def gather_slabs(image, slabs, rank, dst=0, group=None):
    """image: [h, w] tensor (int32 view of the u32 pixels) on every rank; each rank has rendered
    rows of its own slab.  After the call rank `dst` holds the full frame.  Row ranges are
    contiguous in memory, so every peer sends its slice in place and `dst` receives straight into
    the final image: one grouped send/recv, ragged sizes, no staging copy."""
    import torch.distributed as dist
    h = image.shape[0]
    ops = []
    if rank == dst:
        for r, s in enumerate(slabs):
            a, b = slab_pixel_rows(s, h)
            if r != dst and b > a:
                ops.append(dist.P2POp(dist.irecv, image[a:b], r, group))
    else:
        a, b = slab_pixel_rows(slabs[rank], h)
        if b > a:
            ops.append(dist.P2POp(dist.isend, image[a:b], dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return image
