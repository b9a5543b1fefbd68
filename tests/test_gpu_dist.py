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
