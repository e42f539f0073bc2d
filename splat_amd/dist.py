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


def slab_partition_balanced(row_loads, world_size, row_overhead=0.0):
    """Contiguous tile-row slabs minimising the heaviest slab (classic linear partition, by bisection
    on the bottleneck).  row_loads: per-tile-row cost estimate (Renderer.tile_row_loads); every rank
    computes the same partition from the same numbers, so nothing is exchanged.  Ranks get at least
    one row while rows last."""
    loads = [float(v) + float(row_overhead) for v in row_loads]
    n = len(loads)
    if world_size >= n:
        return [(min(k, n), min(k + 1, n)) for k in range(world_size)]

    def cuts(limit):
        out, acc, start = [], 0.0, 0
        for i, v in enumerate(loads):
            if acc + v > limit and i > start:
                out.append((start, i))
                start, acc = i, 0.0
            acc += v
        out.append((start, n))
        return out

    lo, hi = max(loads), sum(loads)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if len(cuts(mid)) <= world_size:
            hi = mid
        else:
            lo = mid
    slabs = cuts(hi)
    # fewer slabs than ranks: split the longest ones so that every rank has rows
    while len(slabs) < world_size:
        k = max(range(len(slabs)), key=lambda j: (slabs[j][1] - slabs[j][0] > 1, sum(loads[slabs[j][0]:slabs[j][1]])))
        a, b = slabs[k]
        if b - a < 2:
            break
        acc, half, cut = 0.0, 0.5 * sum(loads[a:b]), a + 1
        for i in range(a, b - 1):
            acc += loads[i]
            cut = i + 1
            if acc >= half:
                break
        slabs[k:k + 1] = [(a, cut), (cut, b)]
    while len(slabs) < world_size:
        slabs.append((n, n))
    return slabs


def slab_pixel_rows(slab, height):
    return min(slab[0] * TILE, int(height)), min(slab[1] * TILE, int(height))


def gather_slabs(image, slabs, rank, dst=0, group=None):
    """image: [h, w] tensor (int32 view of the u32 pixels) on every rank; each rank has rendered
    rows of its own slab.  After the call rank `dst` holds the full frame.  Row ranges are
    contiguous in memory, so every peer sends its slice in place and `dst` receives straight into
    the final image: one grouped send/recv, ragged sizes, no staging copy."""
    import torch.distributed as dist
    # the op list only depends on the image buffer and the partition: build it once per (buffer,
    # partition) -- at 8 ranks a frame is ~0.2 ms and rank 0's Python time per frame counts
    key = (image.data_ptr(), tuple(image.shape), tuple(map(tuple, slabs)), rank, dst, id(group))
    ops = _GATHER_OPS.get(key)
    if ops is None:
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
        if len(_GATHER_OPS) > 64:
            _GATHER_OPS.clear()
        _GATHER_OPS[key] = ops
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return image


_GATHER_OPS = {}
