"""Row-band partition + the single gather (N > 1 path), run over gloo on the CPU with
world_size 2 and 3: bands cut from a golden frame assemble bit-identically on rank 0."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.npz")


def test_band_rows_tile_the_frame(rwr):
    from rwr_amd.partition import band_rows
    for h in (1080, 2160, 54, 7, 1):
        for n in (1, 2, 3, 4, 8):
            bands = [band_rows(r, n, h) for r in range(n)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(bands[i][1] == bands[i + 1][0] for i in range(n - 1))
            sizes = [b - a for a, b in bands]
            assert max(sizes) - min(sizes) <= 1
    assert band_rows(3, 8, 1080) == (405, 540) and band_rows(7, 8, 2160) == (1890, 2160)   # 135 / 270 rows
    with pytest.raises(ValueError):
        band_rows(2, 2, 10)


def test_strip_rows_tile_the_frame(rwr):
    """The interleaved partition: every row belongs to exactly one rank, strips are 8 rows, shares differ by one strip."""
    import rwr_amd.partition as part
    for height in (1080, 2160, 70, 7, 8, 9, 1):
        for world in (1, 2, 3, 8):
            owner = np.full(height, -1)
            sizes = []
            for r in range(world):
                rows = part.strip_rows(r, world, height)
                assert all(owner[y] == -1 for y in rows)
                owner[rows] = r
                sizes.append(len(rows))
                assert all((y // part.STRIP_ROWS) % world == r for y in rows)
            assert (owner >= 0).all()
            assert max(sizes) - min(sizes) <= part.STRIP_ROWS
    with pytest.raises(ValueError):
        part.strip_rows(2, 2, 100)


def test_library_band_partition_is_the_same(rwr):
    """rwr_dist_band (what rwr_dist_gather_rgba8 uses for every rank's band) == partition.band_rows (what the gloo
    rehearsal below assembles with)."""
    from rwr_amd.partition import band_rows
    for h in (1080, 2160, 54, 7, 1, 4321):
        for n in (1, 2, 3, 4, 5, 8):
            for r in range(n):
                assert rwr.dist_band(r, n, h) == band_rows(r, n, h)
    with pytest.raises(rwr.RwrError):
        rwr.dist_band(2, 2, 10)


def _worker(rank, world, port, height_cut, result_path):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    graft.load_package()
    from rwr_amd.partition import band_rows, gather_bands_equal, gather_bands_ragged, make_gather_list

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(GOLDEN, allow_pickle=False)
    full = z["suzanne_oblique_spheres/color"][:height_cut]          # (h, w, 4) uint8 — what each rank would render
    h, w = full.shape[:2]
    r0, r1 = band_rows(rank, world, h)
    band = torch.from_numpy(np.ascontiguousarray(full[r0:r1]).reshape(-1).copy())
    frame = torch.zeros(h * w * 4, dtype=torch.uint8) if rank == 0 else None
    gl = make_gather_list(frame, world, w, h, 4) if rank == 0 else None
    gather = gather_bands_equal if h % world == 0 else gather_bands_ragged
    for _ in range(3):                                               # repeated frames reuse the same buffers
        gather(dist, band, gl, rank, 0)
    dist.barrier()
    if rank == 0:
        np.save(result_path, frame.numpy().reshape(h, w, 4))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height_cut", [(2, 60), (3, 60), (2, 59), (3, 58)])
def test_gather_assembles_frame_over_gloo(rwr, tmp_path, world, height_cut):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() * 7 + world * 13 + height_cut) % 2000
    result = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, port, height_cut, result), nprocs=world, join=True)
    got = np.load(result)
    want = np.load(GOLDEN, allow_pickle=False)["suzanne_oblique_spheres/color"][:height_cut]
    assert np.array_equal(got, want)
