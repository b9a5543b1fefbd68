"""Frame partition across GPUs: N contiguous row bands — or every N-th strip of 8 rows — one process
per GPU, and ONE gather of the finished bands to rank 0 (RCCL over xGMI on the GPU
box; the same code runs over gloo in the CPU tests).  Pixels are independent, the
scene is replicated, and ranks address pixels / key the RNG by GLOBAL coordinates,
so the assembled frame is bit-identical for every N."""
from __future__ import annotations


def band_rows(rank: int, world: int, height: int) -> tuple[int, int]:
    """Row band [begin, end) of `rank`; bands tile [0, height) exactly, sizes differ by <= 1."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (rank * height) // world, ((rank + 1) * height) // world


STRIP_ROWS = 8   # RWR_STRIP_ROWS: rows of a strip = of every render kernel's workgroup tile


def strip_rows(rank: int, world: int, height: int) -> list[int]:
    """Rows of the INTERLEAVED partition (rwr_render_strips(ctx, ..., rank, world)): strips rank, rank + world, ... of 8 rows.
    Every rank gets the same share of whatever part of the screen the scene covers."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return [y for s in range(rank, (height + STRIP_ROWS - 1) // STRIP_ROWS, world)
            for y in range(s * STRIP_ROWS, min(height, (s + 1) * STRIP_ROWS))]


def make_gather_list(frame_flat, world: int, width: int, height: int, bytes_per_pixel: int):
    """Views into rank 0's full-frame byte tensor, one per rank, in final image order."""
    out = []
    for r in range(world):
        a, b = band_rows(r, world, height)
        out.append(frame_flat[a * width * bytes_per_pixel:b * width * bytes_per_pixel])
    return out


def gather_bands_equal(dist, band, gather_list, rank: int, dst: int = 0):
    """The single collective of a frame when height % world == 0 (1080 and 2160 rows split
    evenly over 1/2/4/8 GPUs): every rank contributes its band, rank `dst` receives all of
    them directly in place, in final image order."""
    dist.gather(band, gather_list if rank == dst else None, dst=dst)


def gather_bands_ragged(dist, band, gather_list, rank: int, dst: int = 0):
    """Same exchange when bands differ by a row (gather needs equal sizes): point-to-point."""
    world = dist.get_world_size()
    if rank == dst:
        gather_list[dst].copy_(band)
        reqs = [dist.irecv(gather_list[r], src=r) for r in range(world) if r != dst]
        for q in reqs:
            q.wait()
    else:
        dist.send(band, dst=dst)
