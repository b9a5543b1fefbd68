"""Partition of a frame across ranks + the layout of its single gather (the N > 1 path), on the CPU: the library's own
layout function (rwr_dist_strip_layout — the arithmetic rwr_dist_gather_strips_rgba8 sizes its sends, receives and receive
offsets with, csrc/rwr_strips.h) against an independent statement of the partition, and the gather itself over gloo with
world sizes 2 and 3: every rank packs its message with the library, the root receives at the library's offsets and deals
out with the library — a golden frame comes back byte for byte.  (The device forms of pack / deal-out and the exchange's
stand-in run in tests/test_gpu_dist.py.)"""
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


def test_library_strip_layout_matches_the_partition(rwr):
    """rwr_dist_strip_layout (what the gather uses) against partition.strip_rows (written independently): strip and row
    counts, the tail owner, receive offsets = whole strips of the ranks before; worlds larger than the strip count too."""
    import rwr_amd.partition as part
    for height in (1080, 2160, 180, 181, 67, 70, 20, 9, 8, 7, 1):
        n_strips = (height + 7) // 8
        for world in (1, 2, 3, 4, 5, 8, 16):
            at = 0
            for r in range(world):
                lay = rwr.dist_strip_layout(r, world, height)
                rows = part.strip_rows(r, world, height)
                assert lay["n_strips"] == n_strips and lay["recv_rows_total"] == 8 * n_strips
                assert lay["rows"] == len(rows)
                assert lay["strips"] == len(range(r, n_strips, world))
                assert lay["owns_tail"] == int(height % 8 != 0 and (n_strips - 1) % world == r)
                assert lay["recv_row"] == at
                at += 8 * lay["strips"]
            assert at == 8 * n_strips
    with pytest.raises(rwr.RwrError):
        rwr.dist_strip_layout(2, 2, 10)


@pytest.mark.parametrize("height", [60, 59, 20, 7])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_host_pack_and_deal_round_trip(rwr, world, height):
    """Every rank's message packed from a frame, laid side by side at the library's receive offsets, dealt out: the frame."""
    import rwr_amd.partition as part
    frame = np.load(GOLDEN, allow_pickle=False)["suzanne_oblique_spheres/color"][:height]
    w = frame.shape[1]
    recv = np.full((rwr.dist_strip_layout(0, world, height)["recv_rows_total"], w, 4), 0xA5, np.uint8)
    for r in range(world):
        lay = rwr.dist_strip_layout(r, world, height)
        msg = rwr.dist_host_pack_strips(r, world, frame)
        assert np.array_equal(msg, frame[part.strip_rows(r, world, height)])     # the rank's rows, ascending
        recv[lay["recv_row"]:lay["recv_row"] + lay["rows"]] = msg
    assert np.array_equal(rwr.dist_host_deal_strips(world, recv, height), frame)


def _worker(rank, world, port, height_cut, strips, result_path):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import __graft_entry__ as graft
    rwr = graft.load_package()

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(GOLDEN, allow_pickle=False)
    full = z["suzanne_oblique_spheres/color"][:height_cut]          # (h, w, 4) uint8 — what the ranks would render between them
    h, w = full.shape[:2]
    # this rank's frame: only ITS share is rendered, the rest is whatever the slot held before (made wrong on purpose)
    mine = np.full_like(full, 17 + rank)
    if strips:
        from rwr_amd.partition import strip_rows
        rows = strip_rows(rank, world, h)
        mine[rows] = full[rows]
        send = torch.from_numpy(rwr.dist_host_pack_strips(rank, world, mine).reshape(-1).copy())   # the library's pack
        spans = [(lay["recv_row"], lay["rows"]) for lay in (rwr.dist_strip_layout(r, world, h) for r in range(world))]
        recv_rows = rwr.dist_strip_layout(0, world, h)["recv_rows_total"]
    else:
        a, b = rwr.dist_band(rank, world, h)
        mine[a:b] = full[a:b]
        send = torch.from_numpy(mine[a:b].reshape(-1).copy())
        spans = [(ra, rb - ra) for ra, rb in (rwr.dist_band(r, world, h) for r in range(world))]   # bands land in place
        recv_rows = h
    recv = torch.zeros(recv_rows * w * 4, dtype=torch.uint8) if rank == 0 else None
    for _ in range(3):                                               # repeated frames reuse the same buffers
        # the frame's ONE exchange: every rank's message to the root, received at the library's offsets (gloo send/recv here,
        # one grouped ncclSend/ncclRecv on the GPU box)
        if rank == 0:
            at, n = spans[0]
            recv[at * w * 4:(at + n) * w * 4].copy_(send)
            reqs = [dist.irecv(recv[spans[r][0] * w * 4:(spans[r][0] + spans[r][1]) * w * 4], src=r) for r in range(1, world) if spans[r][1]]
            for q in reqs:
                q.wait()
        elif send.numel():
            dist.send(send, dst=0)
    dist.barrier()
    if rank == 0:
        got = recv.numpy().reshape(recv_rows, w, 4)
        np.save(result_path, rwr.dist_host_deal_strips(world, got, h) if strips else got)   # the library's deal-out
    dist.destroy_process_group()


@pytest.mark.parametrize("strips", [True, False], ids=["strips", "bands"])
@pytest.mark.parametrize("world,height_cut", [(2, 60), (3, 60), (2, 59), (3, 58), (3, 13)])
def test_gather_assembles_frame_over_gloo(rwr, tmp_path, world, height_cut, strips):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() * 7 + world * 13 + height_cut + 101 * strips) % 2000
    result = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, port, height_cut, strips, result), nprocs=world, join=True)
    got = np.load(result)
    want = np.load(GOLDEN, allow_pickle=False)["suzanne_oblique_spheres/color"][:height_cut]
    assert np.array_equal(got, want)
