"""GPU parity for the fused primary-ray kernel (through the C ABI) against the CPU
oracle — the reference's frame: clear, two sphere passes, depth copies, mesh pass
(/root/reference/src/lib.rs:1024-1184).

Bars (DESIGN.md "Numerics"): object id, hit distance and depth are BIT-EXACT
(integer/index work and the IEEE-defined float chain that decides visibility);
float colour within 1e-4 absolute per channel (north_star's tolerance; the only
non-IEEE step is pow(x,32)); RGBA8 within 1 LSB.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COLOR_TOL = 1e-4


def _render_gpu(rwr, ctx, model, spheres, cam_inv, w, h, flags=0, rows=None):
    ctx.upload_model(model)
    ctx.set_instances(None)
    ctx.set_spheres(spheres)
    ctx.resize(w, h)
    ctx.render(cam_inv, rwr.make_params(flags=rwr.FLAG_AUX_OUTPUTS | flags), rows=rows)
    return ctx.readback(aux=True)


def _assert_parity(got, want):
    assert np.array_equal(got["obj_id"], want["obj_id"])
    assert np.array_equal(got["hit_t"].view(np.uint32), want["hit_t"].view(np.uint32))
    assert np.array_equal(got["depth"].view(np.uint32), want["depth"].view(np.uint32))
    err = np.abs(got["color_f32"] - want["color_f32"]).max()
    assert err <= COLOR_TOL, f"max colour error {err}"
    d8 = np.abs(got["color"].astype(np.int32) - want["color"].astype(np.int32))
    assert d8.max() <= 1
    assert (d8 != 0).mean() < 1e-3


CAMERAS = {
    "reference_default_inside_mesh": dict(eye=(0, 0, 0), target=(0, 0, -1)),   # lib.rs:352-360
    "s_x15_outside": dict(eye=(0, 0, 3), target=(0, 0, -1)),                     # 15 x 'S' 
    "oblique": dict(eye=(2.5, 1.0, 2.0), target=(0, 0, 0)),
    "spheres_visible": dict(eye=(0.5, 0.4, 0.5), target=(0.5, 0.45, -3.5)),
    "far_away": dict(eye=(0, 0, 40), target=(0, 0, -1)),
}


@pytest.mark.parametrize("cam_name", sorted(CAMERAS))
@pytest.mark.parametrize("size", [(192, 108), (67, 45)])  # second one: ragged tiles on both axes
def test_suzanne_matches_oracle(rwr, orc, gpu_ctx, suzanne, cam_name, size):
    w, h = size
    cam = rwr.make_camera(aspect=w / h, **CAMERAS[cam_name])
    cam_inv = rwr.camera_build_inv_uniform(cam)
    got = _render_gpu(rwr, gpu_ctx, suzanne, rwr.make_spheres(), cam_inv, w, h)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), suzanne)
    _assert_parity(got, want)


def test_cube_config1_matches_oracle(rwr, orc, gpu_ctx, cube):
    """BASELINE.json configs[0]: cube.obj, 256x256, reference camera."""
    w = h = 256
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=1.0))
    got = _render_gpu(rwr, gpu_ctx, cube, rwr.make_spheres(), cam_inv, w, h)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), cube)
    _assert_parity(got, want)
    assert (got["obj_id"] >= 0).all()           # 65 536 / 65 536 mesh pixels
    assert (got["color"] == 255).all()          # Ka = 1.0 saturates every channel


def test_cube_outside_view(rwr, orc, gpu_ctx, cube):
    w, h = 160, 120
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(2.2, 1.7, 3.1), target=(0, 0, 0), aspect=w / h))
    got = _render_gpu(rwr, gpu_ctx, cube, rwr.make_spheres(), cam_inv, w, h)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), cube)
    _assert_parity(got, want)


def test_culling_is_invisible(rwr, gpu_ctx, suzanne, cube):
    """Tile-frustum culling may only skip faces no ray of the tile can hit: every
    output plane must be bit-identical with the brute-force loop."""
    for model, cam_kw, (w, h) in [(suzanne, CAMERAS["s_x15_outside"], (320, 180)),
                                  (suzanne, CAMERAS["reference_default_inside_mesh"], (320, 180)),
                                  (cube, dict(eye=(2.2, 1.7, 3.1), target=(0, 0, 0)), (200, 200)),
                                  (cube, dict(eye=(0, 0, 0), target=(0, 0, -1)), (256, 256))]:
        cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cam_kw))
        a = _render_gpu(rwr, gpu_ctx, model, rwr.make_spheres(), cam_inv, w, h)
        b = _render_gpu(rwr, gpu_ctx, model, rwr.make_spheres(), cam_inv, w, h, flags=rwr.FLAG_NO_CULL)
        for k in a:
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k


def test_spheres_only_and_empty_scene(rwr, orc, gpu_ctx, suzanne):
    w, h = 128, 96
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h))
    empty = dict(suzanne, faces=suzanne["faces"][:0])
    got = _render_gpu(rwr, gpu_ctx, empty, rwr.make_spheres(), cam_inv, w, h)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), empty)
    _assert_parity(got, want)
    assert (got["obj_id"] == -3).any()  # the front sphere (lib.rs:534) covers the rear one from here
    far = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(3, 0.45, -3.5), target=(0.5, 0.45, -3.5), aspect=w / h))
    got = _render_gpu(rwr, gpu_ctx, empty, rwr.make_spheres(), far, w, h)
    want = orc.render_frame(far.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), empty)
    _assert_parity(got, want)
    assert (got["obj_id"] == -2).any() and (got["obj_id"] == -3).any()
    cam_inv = far
    # nothing at all: every pixel keeps the clear value
    got = _render_gpu(rwr, gpu_ctx, empty, rwr.make_spheres([]), cam_inv, w, h)
    assert not got["color"].any() and not got["depth"].any() and (got["obj_id"] == -1).all()


def test_row_bands_assemble_bit_identically(rwr, gpu_ctx, suzanne):
    """Multi-GPU contract: bands rendered separately equal the full frame."""
    w, h = 200, 120
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(eye=(0, 0, 3), aspect=w / h))
    full = _render_gpu(rwr, gpu_ctx, suzanne, rwr.make_spheres(), cam_inv, w, h)
    for n in (2, 3, 8):
        edges = [round(i * h / n) for i in range(n + 1)]
        assembled = {k: np.zeros_like(v) for k, v in full.items()}
        for r0, r1 in zip(edges[:-1], edges[1:]):
            part = _render_gpu(rwr, gpu_ctx, suzanne, rwr.make_spheres(), cam_inv, w, h, rows=(r0, r1))
            for k in assembled:
                assembled[k][r0:r1] = part[k][r0:r1]
        for k in full:
            assert np.array_equal(full[k].view(np.uint8), assembled[k].view(np.uint8)), (n, k)


def test_full_size_config2_properties(rwr, orc, gpu_ctx, suzanne):
    """BASELINE.json configs[1] at full size (1920x1080): SURVEY §8(c) known
    answers + a strided comparison with the oracle (every 8th row)."""
    w, h = 1920, 1080
    cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h))
    got = _render_gpu(rwr, gpu_ctx, suzanne, rwr.make_spheres(), cam_inv, w, h)
    # camera sits inside the mesh: every pixel shows a face (bar one f32 crack pixel)
    assert (got["obj_id"] >= 0).sum() >= w * h - 4
    assert got["obj_id"][540, 960] == 40 and abs(got["hit_t"][540, 960] - 0.708794) < 1e-6
    assert got["obj_id"][100, 100] == 95 and got["obj_id"][900, 1800] == 40
    np.testing.assert_allclose(got["color_f32"][540, 960, :3], (0.1375972, 0.0674279, 0.0622155), atol=2e-6)
    mean = np.clip(got["color_f32"][..., :3], 0, 1).reshape(-1, 3).mean(0)
    np.testing.assert_allclose(mean, (0.12034, 0.07311, 0.06953), atol=2e-5)
    want = orc.render_frame(cam_inv.view(orc.CAMERA_INV_DTYPE), orc.make_screen(w, h), orc.make_spheres(), suzanne)
    _assert_parity(got, want)


def test_two_pixel_and_one_pixel_kernels_agree(rwr, gpu_ctx, suzanne, cube):
    """k_primary_p2 (two pixels per lane, packed f32) and k_primary (one pixel per lane) run the same
    operation sequence per pixel: every plane must be bit-identical, odd widths included."""
    cases = [(suzanne, CAMERAS["reference_default_inside_mesh"], (321, 97)), (suzanne, CAMERAS["s_x15_outside"], (130, 67)),
             (suzanne, CAMERAS["spheres_visible"], (64, 64)), (cube, dict(eye=(2.2, 1.7, 3.1), target=(0, 0, 0)), (99, 50)),
             (cube, dict(eye=(0, 0, 0), target=(0, 0, -1)), (1, 9)), (suzanne, CAMERAS["oblique"], (15, 3))]
    for model, cam_kw, (w, h) in cases:
        cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cam_kw))
        a = _render_gpu(rwr, gpu_ctx, model, rwr.make_spheres(), cam_inv, w, h)
        b = _render_gpu(rwr, gpu_ctx, model, rwr.make_spheres(), cam_inv, w, h, flags=rwr.FLAG_ONE_PIXEL_PER_LANE)
        for k in a:
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (w, h, k)


def test_bvh_frame_kernel_agrees(rwr, gpu_ctx, suzanne, cube):
    """RWR_FLAG_USE_BVH (per-ray BVH traversal for the mesh pass) is bit-identical to the
    candidate-list kernel: same exact test, ties by face index."""
    for model, cam_kw, (w, h) in [(suzanne, CAMERAS["reference_default_inside_mesh"], (200, 113)), (suzanne, CAMERAS["far_away"], (96, 54)),
                                  (suzanne, CAMERAS["spheres_visible"], (77, 64)), (cube, dict(eye=(2.2, 1.7, 3.1), target=(0, 0, 0)), (128, 96))]:
        cam_inv = rwr.camera_build_inv_uniform(rwr.make_camera(aspect=w / h, **cam_kw))
        a = _render_gpu(rwr, gpu_ctx, model, rwr.make_spheres(), cam_inv, w, h)
        b = _render_gpu(rwr, gpu_ctx, model, rwr.make_spheres(), cam_inv, w, h, flags=rwr.FLAG_USE_BVH)
        for k in a:
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (w, h, k)
