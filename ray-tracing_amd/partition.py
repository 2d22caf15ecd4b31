"""Sample-index partition for multi-GPU renders.

Every sample of the reference's kernel is a pure function of (pixel, currentSample, rayIndex): its seed is
re-derived from those three integers (reference: "Ray Tracer/Shaders.metal":121-126) and nothing carries
over between samples.  Splitting the rayIndex range [0, spp) into disjoint contiguous ranges, one per rank,
therefore produces exactly the same multiset of sample colours as one GPU tracing all of them; only the
order of the float additions differs.  The ranks' linear float4 sums are then added (RCCL reduce).
"""


def ray_range(rank, world_size, spp):
    """Contiguous range [begin, end) of rayIndex values for `rank`; ranges of all ranks tile [0, spp)."""
    if world_size < 1 or not (0 <= rank < world_size) or spp < 0:
        raise ValueError("bad partition arguments")
    base, rem = divmod(spp, world_size)
    begin = rank * base + min(rank, rem)
    end = begin + base + (1 if rank < rem else 0)
    return begin, end


def tile_rows(rank, world_size, height, tile=64):
    """Fallback partition by image rows in units of the reference's 64-pixel tiles: rows [y0, y1) of `rank`."""
    ntiles = (height + tile - 1) // tile
    b, e = ray_range(rank, world_size, ntiles)
    return min(b * tile, height), min(e * tile, height)
