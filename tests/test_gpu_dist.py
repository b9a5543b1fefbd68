"""The multi-GPU path of the C ABI (rwr_dist_*): one process per GPU, interleaved strips (or row bands), ONE RCCL gather
per frame issued by the library itself.  A one-GPU box can only run an RCCL world of one rank — that exercises the whole
call sequence on hardware (RCCL loaded at run time, communicator, grouped send/recv on the frame's stream).  The N > 1
code — message layout, pack launch, receive offsets, deal-out launch — runs here through the library's LOOPBACK form
(rwr_dist_loopback_*): the context plays ranks 0..N-1 in turn and only the ncclSend/ncclRecv pair is replaced by a device
copy to the same address.  The same layout over a real multi-process exchange (gloo): tests/test_partition.py."""
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


@pytest.mark.parametrize("frames_in_flight", [1, 2, 3])
def test_world_of_one_strip_gather_with_frames_in_flight(rwr, suzanne, frames_in_flight):
    """Every frame slot owns its gather set: gathers of consecutive frames overlap the next render, each frame still arrives
    whole (checked frame by frame against a plain render of the same camera)."""
    w, h = 328, 181
    cams = [rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.1 * f, 0.05 * f, 3), aspect=w / h)) for f in range(6)]
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        want = []
        for cam in cams:
            ctx.render(cam, rwr.make_params())
            want.append(ctx.readback()["color"])
        ctx.set_frames_in_flight(frames_in_flight)
        ctx.dist_init(0, 1, rwr.dist_get_unique_id())
        gather = ctx.dist_gather_call(0, strips=True)
        for upto in range(1, len(cams) + 1):       # 1, 2, ... frames enqueued back to back, then the last one read
            for cam in cams[:upto]:
                ctx.render(cam, rwr.make_params(), strips=(0, 1))
                gather()
            assert np.array_equal(ctx.dist_readback(), want[upto - 1]), upto
        ctx.dist_destroy()


@pytest.mark.parametrize("strips", [True, False], ids=["strips", "bands"])
@pytest.mark.parametrize("height", [180, 67, 181, 20])   # 67, 181: short last strip; 20: fewer strips than ranks at N = 8
@pytest.mark.parametrize("frames_in_flight", [1, 2])
def test_loopback_gather_assembles_the_frame(rwr, suzanne, height, strips, frames_in_flight):
    """The library's own gather stages for N = 2, 3, 8 on one GPU: per rank its render of its share, the library's pack launch
    and message size, a device copy to the library's receive offset (in place of Send/Recv); then the library's deal-out.  The
    assembled frame is the whole frame's bytes.  A width that is not a multiple of 4 pixels makes the short strip's byte count
    odd in 16-byte units (the copy kernels' dword tail)."""
    w, h = 203, height
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0.3, 0.2, 2.6), aspect=w / h))
    params = rwr.make_params()
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.resize(w, h)
        ctx.set_frames_in_flight(frames_in_flight)
        with pytest.raises(rwr.RwrError):
            ctx.dist_loopback_finish(2, strips)          # nothing deposited yet
        ctx.render(cam_inv, params)
        full = ctx.readback()["color"]
        assert full.any()
        for n in (2, 3, 8):
            # poison every slot's frame so that a rank's rows can only come from its own render
            other = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(-1.0, 0.4, 2.0), aspect=w / h))
            for _ in range(frames_in_flight):
                ctx.render(other, params)
            for r in range(n):
                if strips:
                    ctx.render(cam_inv, params, strips=(r, n))
                else:
                    ctx.render(cam_inv, params, rows=rwr.dist_band(r, n, h))
                ctx.dist_loopback_deposit(r, n, strips)
                with pytest.raises(rwr.RwrError):
                    ctx.dist_readback()                  # not assembled until the root's side has run
            ctx.dist_loopback_finish(n, strips)
            got = ctx.dist_readback()
            assert np.array_equal(got, full), (n, np.argwhere((got != full).any(axis=(1, 2)))[:8].ravel())
        with pytest.raises(rwr.RwrError):
            ctx.dist_loopback_deposit(3, 3, strips)


def test_loopback_gather_of_configs4_at_full_size(rwr, suzanne):
    """BASELINE configs[4] as its 8 ranks would run it — suzanne x16, 3840x2160, 64 spp + 1 bounce, every 8th strip per rank —
    gathered by the library's stages on one GPU: byte for byte the frame one GPU renders alone."""
    w, h, n = 3840, 2160, 8
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0, 0, 12), aspect=w / h))
    params = rwr.make_params(spp=64, max_bounces=1)
    with rwr.Context(0) as ctx:
        ctx.upload_model(suzanne)
        ctx.set_spheres(rwr.make_spheres())
        ctx.set_instances(rwr.make_instance_grid(4, 3.0))
        ctx.resize(w, h)
        ctx.render(cam_inv, params)
        full = ctx.readback()["color"]
        segments = sum(ctx.last_render_stats())
        ctx.render(rwr.camera_build_inv_uniform(rwr.make_camera(eye=(1, 0, 9), aspect=w / h)), rwr.make_params())   # poison
        got_segments = 0
        for r in range(n):
            ctx.render(cam_inv, params, strips=(r, n))
            got_segments += sum(ctx.last_render_stats())
            ctx.dist_loopback_deposit(r, n, True)
        ctx.dist_loopback_finish(n, True)
        got = ctx.dist_readback()
        assert got_segments == segments
        assert np.array_equal(got, full)
