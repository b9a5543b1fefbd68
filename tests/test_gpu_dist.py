"""The multi-GPU path of the C ABI (rwr_dist_*): one process per GPU, row bands, ONE RCCL gather per frame issued
by the library itself.  A one-GPU box can only run a world of one rank — that still exercises the whole call
sequence on hardware (RCCL loaded at run time, communicator, grouped send/recv on the frame's stream, receive buffer
in final image order); the band arithmetic and N > 1 assembly are covered by tests/test_partition.py over gloo and by
test_gpu_primary.py::test_row_bands_assemble_bit_identically."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frames_in_flight", [1, 2])
def test_world_of_one_gather_equals_readback(rwr, suzanne, frames_in_flight):
    w, h = 320, 180
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0, 0, 3), aspect=w / h))
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        ctx.set_frames_in_flight(frames_in_flight)
        ctx.dist_init(0, 1, rwr.dist_get_unique_id())
        with pytest.raises(rwr.RwrError):
            ctx.dist_readback()                      # nothing gathered yet
        r0, r1 = rwr.dist_band(0, 1, h)
        assert (r0, r1) == (0, h)
        for frame in range(5):                       # several frames: slots alternate, gathers are ordered
            cam = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.1 * frame, 0, 3), aspect=w / h))
            ctx.render(cam, rwr.make_params(), rows=(r0, r1))
            ctx.dist_gather(0)
        got = ctx.dist_readback()
        want = ctx.readback()["color"]
        assert got.any() and np.array_equal(got, want)
        ctx.dist_barrier()
        with pytest.raises(rwr.RwrError):
            ctx.dist_gather(3)                       # root outside the world
        ctx.dist_destroy()
        with pytest.raises(rwr.RwrError):
            ctx.dist_gather(0)                       # no communicator any more


@pytest.mark.parametrize("height", [180, 67])   # 67: the frame's last strip is short
def test_interleaved_strips_assemble_bit_identically(rwr, suzanne, height):
    """rwr_render_strips: rank r of N renders strips r, r + N, ... of 8 rows.  Assembled from N separate renders the frame is
    the whole frame's bytes — the reference frame (all planes) and a path-traced one — for N = 2, 3, 8."""
    import rwr_amd.partition as part
    w, h = 200, height
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.3, 0.2, 2.6), aspect=w / h))
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        for params in (rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS), rwr.make_params(spp=5, max_bounces=1, seed=4, flags=rwr.FLAG_AUX_OUTPUTS)):
            ctx.render(cam_inv, params)
            full = ctx.readback(aux=True)
            primary_full, bounce_full = ctx.last_render_stats()
            for n in (2, 3, 8):
                asm = {k: np.zeros_like(v) for k, v in full.items()}
                primary = bounce = 0
                for r in range(n):
                    ctx.render(cam_inv, params, strips=(r, n))
                    got = ctx.readback(aux=True)
                    a, b = ctx.last_render_stats()
                    primary += a; bounce += b
                    rows = part.strip_rows(r, n, h)
                    for k in asm:
                        asm[k][rows] = got[k][rows]
                for k in asm:
                    assert np.array_equal(asm[k].view(np.uint8), full[k].view(np.uint8)), (n, k)
                assert (primary, bounce) == (primary_full, bounce_full)
        for params in (rwr.make_params(), rwr.make_params(spp=3, max_bounces=1)):
            ctx.render(cam_inv, params, strips=(h // 8 + 3, h // 8 + 5))  # a rank beyond the frame's last strip renders nothing
            ctx.synchronize()
            assert ctx.last_render_stats() == (0, 0)
        with pytest.raises(rwr.RwrError):
            ctx.render(cam_inv, rwr.make_params(), strips=(2, 2))     # first strip must be below the stride
        with pytest.raises(rwr.RwrError):
            ctx.render(cam_inv, rwr.make_params(), strips=(0, 0))


def test_world_of_one_strip_gather_equals_readback(rwr, suzanne):
    """rwr_dist_gather_strips_rgba8 with one rank: pack, exchange with itself, deal out — the frame comes back whole."""
    w, h = 320, 181
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0, 0, 3), aspect=w / h))
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        ctx.dist_init(0, 1, rwr.dist_get_unique_id())
        gather = ctx.dist_gather_call(0, strips=True)
        for frame in range(3):
            cam = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.1 * frame, 0, 3), aspect=w / h))
            ctx.render(cam, rwr.make_params(), strips=(0, 1))
            gather()
        got = ctx.dist_readback()
        want = ctx.readback()["color"]
        assert got.any() and np.array_equal(got, want)
        ctx.dist_destroy()
