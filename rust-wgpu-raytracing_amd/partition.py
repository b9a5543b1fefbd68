"""Frame partition across GPUs, stated independently of the library: N contiguous row bands — or every
N-th strip of 8 rows — one process per GPU.  The gather itself (message layout, receive offsets, pack, deal-out, the RCCL
exchange) lives in librwr_hip.so (rwr_dist_*, csrc/rwr_strips.h); tests/test_partition.py checks the library's layout
against these functions.  Pixels are independent, the scene is replicated, and ranks address pixels / key the RNG by
GLOBAL coordinates, so the assembled frame is bit-identical for every N."""
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
